"""The C++ host mirror end to end: examples/layouts (built by __graft_entry__.build()) renders two of the reference's
layouts as a stand-alone program; its pictures must equal what the same scenes give through the Python harness."""
import os
import subprocess

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "layouts")


def _read_ppm(path):
    raw = open(path, "rb").read()
    assert raw[:3] == b"P6\n"
    parts = raw.split(b"\n", 3)
    w, h = (int(x) for x in parts[1].split())
    assert parts[2] == b"255"
    return np.frombuffer(parts[3], dtype=np.uint8).reshape(h, w, 3)


def test_layout3_focusing_program(ctx, tmp_path):
    import ray_tracing_amd as rta
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    out = str(tmp_path / "focusing.ppm")
    r = subprocess.run([BIN, "focusing", scenes.ICOSA_OBJ, out, "20", "5"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    img = _read_ppm(out)
    g, _ = scenes.build_icosahedron(material="plastic", light="area10", oracle=False)
    cam = rta.Camera(**dict(scenes.C1_CAMERA, res=20))
    cam.set_seed(1)
    cam.render_focusing(ctx, g, 0.1, background=(200, 200, 200), spp=5)
    assert img.shape == (cam.height, cam.width, 3)
    assert (img[..., ::-1] == cam.image()).all()            # PPM is R,G,B; image() is the reference's B,G,R picture


def test_layout1_apples_program(ctx, tmp_path):
    out = str(tmp_path / "apples.ppm")
    r = subprocess.run([BIN, "apples", out, "12"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    img = _read_ppm(out)
    assert img.shape == (86, 115, 3) and img.std() > 20
    assert int(r.stdout.split(",")[1].split()[0]) > 500      # the texture edges are re-sampled
    # wrong usage and a missing file are reported, not crashed on
    assert subprocess.run([BIN], capture_output=True).returncode == 2
    bad = subprocess.run([BIN, "focusing", "/nonexistent.obj", out], capture_output=True, text=True)
    assert bad.returncode == 1 and "cannot open" in bad.stderr


def test_layout5_wine_program(ctx, tmp_path):
    """Layout5_Wine as the reference writes it (three cups, 5x5 area light, Render_PhotonMapper_Antialiasing) through the C++ host
    mirror; compared with the same run driven through the ctypes harness"""
    import ray_tracing_amd as rta
    out = str(tmp_path / "wine.ppm")
    r = subprocess.run([BIN, "wine", out, "8", "2", "600", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = _read_ppm(out)
    assert "24 objects" in r.stdout
    g, _ = scenes.build_wine(cups="layout", light="area5", oracle=False)
    cam = rta.Camera(view=(-23.5, 27.5, 24), eye=(2.9, -3, -2), left=(-1, -1, 0), dist=15, hor=8, ver=6, res=8, depth=8)
    pm = rta.PhotonMap(ctx, g, cam, photon_num=600, times=2, background=(0, 0, 0), seed=1, aa=rta.AaParams(oversample_times=3))
    for t in range(2):
        pm.run_pass(t)
    pic = pm.picture().reshape(cam.height, cam.width, 3)
    assert img.shape == pic.shape and int(r.stdout.split(",")[2].split()[0]) == int(pm.edge_mask().sum()) > 50
    assert (img == pic.astype(np.uint8)).all()


def test_layout6_room_program(ctx, tmp_path):
    out = str(tmp_path / "room.ppm")
    r = subprocess.run([BIN, "room", scenes.ICOSA_OBJ, out, "10", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    img = _read_ppm(out)
    assert img.shape == (60, 80, 3) and img.std() > 10
    assert int(r.stdout.split(",")[1].split()[0]) > 100


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stderr
    return r


@pytest.mark.parametrize("force_rccl", ["0", "1"])
def test_cpp_host_shards_over_a_communicator(ctx, tmp_path, force_rccl):
    """The C++ host's own sharded renders (rt::host::Camera::Render_RayTracer / Render_Focusing / Render_PhotonMapper over a Comm: the loop of
    Camera.h:59-63 split without any Python): a communicator of ONE rank -- two RCCL ranks cannot share the one GPU of the test box -- once as the
    plain short cut and once with every collective going through RCCL (RT_HIP_COMM_FORCE_RCCL=1); the pictures must equal the unsharded program's
    bit for bit.  (More than one rank: tests/test_gpu_tiles.py reassembles the same tile shards, sample slices and photon ranges through the same
    entry points; an N-GPU run is the driver's.)"""
    env = {"RT_HIP_COMM_FORCE_RCCL": force_rccl}
    for what, args in (("raytracer", [scenes.ICOSA_OBJ, None, "40", "3"]), ("focusing", [scenes.ICOSA_OBJ, None, "20", "4"]), ("wine", [None, "12", "2", "800", "0"])):
        plain, shard = str(tmp_path / (what + "_plain.ppm")), str(tmp_path / (what + "_shard.ppm"))
        idf = str(tmp_path / (what + force_rccl + ".id"))
        _run([what] + [plain if a is None else a for a in args])
        _run([what] + [shard if a is None else a for a in args] + ["--rank", "0", "--world", "1", "--id-file", idf], env=env)
        a, b = _read_ppm(plain), _read_ppm(shard)
        assert a.shape == b.shape and (a == b).all(), what
        assert a.std() > 5, what                                      # a picture, not a constant frame
