"""Digest gpurun_out/prof_pmc_{a,b,c} (tools/collect_profiles.sh) into profiles/<round>_traffic.json: per-kernel, per-launch PMC
averages over the non-statistics launches.  hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM
section: FETCH_SIZE/WRITE_SIZE are in KiB-less raw bytes here as rocprofv3 reports them x 1; gfx950 under-reports the fetch of
wide streams by half)."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
    f = glob.glob(os.path.join(ROOT, "gpurun_out", d, "*counter_collection.csv"))
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "<true" in k or "at::native" in k or "hipcub" in k or "rocprim" in k:
            continue        # statistics builds of the kernels and torch's own kernels
        name = k.split("(")[0].replace("void ", "").split("<")[0]
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[name][row["Counter_Name"]] += 1
    return {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}


a, b, c = load("prof_pmc_a"), load("prof_pmc_b"), load("prof_pmc_c")
out = {}
for k in sorted(set(a) | set(b) | set(c)):
    fa, fb = a.get(k, {}), b.get(k, {})
    fetch, write = fa.get("FETCH_SIZE", 0.0) * 1024.0, fb.get("WRITE_SIZE", 0.0) * 1024.0     # rocprofv3 reports KiB
    out[k] = {"FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write,
              "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide 16 B/lane stream); WRITE_SIZE as is",
              "TCC_HIT": fa.get("TCC_HIT_sum", 0.0), "TCC_MISS": fb.get("TCC_MISS_sum", 0.0), "TCC_REQ": fb.get("TCC_REQ_sum", 0.0), "sq": c.get(k, {})}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print("%-12s hbm %.1f MB/launch  (fetch raw %.1f MB, write %.1f MB)" % (k, v["hbm_bytes_per_launch"] / 1e6, v["FETCH_SIZE_bytes_raw"] / 1e6, v["WRITE_SIZE_bytes"] / 1e6))
