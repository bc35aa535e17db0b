"""Every parity ALLOWANCE of the GPU tests -- a count of rays / pixels / hit points that may differ from the oracle for a named cause -- goes
through allow(): the observed count is written to a JSON file the repository keeps (profiles/r03_parity_counts.json is a copy of a GPU-box
run), and asserted against its bar.  Bars are north_star's (hit ids: ties <= 1e-4 of the rays; colours: within 1e-3 relative) unless a named
cause stands, and then a small multiple of the count observed on the GPU box -- never a blanket fraction.

The file: $RT_PARITY_COUNTS, else <repo>/gpurun_out/parity_counts.json (gpurun merges gpurun_out/ back).  One entry per (test, name)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("RT_PARITY_COUNTS") or os.path.join(ROOT, "gpurun_out", "parity_counts.json")


def _test_name():
    t = os.environ.get("PYTEST_CURRENT_TEST", "")
    return t.split(" ")[0].split("::", 1)[-1] if t else "(no test)"


def record(name, **fields):
    """keep an observation under (current test, name)"""
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        try:
            data = json.load(open(PATH))
        except Exception:
            data = {}
        data.setdefault(_test_name(), {})[name] = fields
        tmp = PATH + ".tmp"
        json.dump(data, open(tmp, "w"), indent=1, sort_keys=True)
        os.replace(tmp, PATH)
    except OSError:
        pass            # a read-only tree must not fail a parity test


def allow(name, count, total, bar=None, bar_count=None, cause=None):
    """`count` of `total` items differ.  bar: allowed fraction of total; bar_count: allowed absolute count (either or both).
    Records the observation and asserts it."""
    count, total = int(count), int(total)
    limit = None
    if bar is not None:
        limit = bar * total
    if bar_count is not None:
        limit = bar_count if limit is None else max(limit, bar_count)
    record(name, count=count, total=total, fraction=(count / total if total else 0.0), bar_fraction=bar, bar_count=bar_count, cause=cause)
    assert limit is not None and count <= limit, "%s: %d of %d differ (allowed %s); cause named: %s" % (name, count, total, limit, cause)
    return count


def value(name, observed, bar, cause=None):
    """a scalar error (max relative colour error, max |dt| ...) against its bar"""
    observed = float(observed)
    record(name, value=observed, bar=bar, cause=cause)
    assert observed < bar, "%s: %g (bar %g)" % (name, observed, bar)
    return observed
