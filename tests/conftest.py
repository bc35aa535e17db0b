import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    import ray_tracing_amd as rta
    if rta.device_count() < 1:
        pytest.fail("no HIP device visible: gpu-marked tests must run on the GPU box (there is no CPU fallback)")
    return rta.Context(0)
