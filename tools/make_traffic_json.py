"""Digest gpurun_out/prof_<tag>/pmc_{a,b,c}_<config> (tools/collect_profiles.sh) into profiles/<tag>_traffic.json: per kernel, per-launch
PMC averages over the non-statistics launches, keyed "kernel" for the default C2 run and "kernel:C3" ... for the other configs, and copy
the kernel-trace statistics to profiles/<tag>_kernel_stats_<config>.csv.
hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section: rocprofv3 reports KiB; gfx950 under-reports the
fetch of wide streams by half, so the doubled figure is an upper bound).  kernel_cycles = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
BASE = os.path.join(ROOT, "gpurun_out", "prof_" + tag)


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.Counter())
    f = glob.glob(os.path.join(BASE, d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return {}
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"]
        if "<true, " in k or "at::native" in k or "hipcub" in k or "rocprim" in k or "elementwise" in k:
            continue        # statistics builds of the walk kernels (<STATS, ...>; single-parameter templates are <PARAM>) and torch's / rocPRIM's own kernels
        name = k.split("(")[0].replace("void ", "").split("<")[0]
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[name][row["Counter_Name"]] += 1
    return {k: dict({c: acc[k][c] / cnt[k][c] for c in acc[k]}, launches=max(cnt[k].values())) for k in acc}


out = {}
for cfg in ("C2", "C3", "C4", "C5"):
    a, b, c = load("pmc_a_" + cfg), load("pmc_b_" + cfg), load("pmc_c_" + cfg)
    d, e5 = load("pmc_d_" + cfg), load("pmc_e_" + cfg)
    for k in sorted(set(a) | set(b) | set(c)):
        fa, fb, fc = a.get(k, {}), b.get(k, {}), c.get(k, {})
        fc = dict(fc); fc.update({x: v for x, v in d.get(k, {}).items() if x != "launches"}); fc.update({x: v for x, v in e5.get(k, {}).items() if x != "launches"})
        fetch, write = fa.get("FETCH_SIZE", 0.0) * 1024.0, fb.get("WRITE_SIZE", 0.0) * 1024.0
        e = {"launches_averaged": fc.get("launches", fa.get("launches", 0)), "FETCH_SIZE_bytes_raw": fetch, "WRITE_SIZE_bytes": write, "hbm_bytes_per_launch": 2 * fetch + write,
             "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide 16 B/lane stream); WRITE_SIZE as is",
             "TCC_HIT": fa.get("TCC_HIT_sum", 0.0), "TCC_MISS": fb.get("TCC_MISS_sum", 0.0), "TCC_REQ": fb.get("TCC_REQ_sum", 0.0),
             "sq": {x: v for x, v in fc.items() if x != "launches"}}
        if "SQ_INSTS_VALU" in fc:
            e["SQ_INSTS_VALU"] = fc["SQ_INSTS_VALU"]
        f64 = sum(fc.get("SQ_INSTS_VALU_%s_F64" % t, 0.0) for t in ("ADD", "MUL", "FMA", "TRANS"))
        if "SQ_INSTS_VALU" in fc and "SQ_INSTS_VALU_FMA_F64" in fc and fc["SQ_INSTS_VALU"] > 0:
            e["valu_f64_share"] = f64 / fc["SQ_INSTS_VALU"]          # the instructions that take 4 cycles per wave64 instruction; everything else 2
        if "GRBM_GUI_ACTIVE" in fc:
            e["kernel_cycles"] = fc["GRBM_GUI_ACTIVE"] / 8.0
            if "SQ_INSTS_VALU" in fc and fc["GRBM_GUI_ACTIVE"] > 0:
                e["valu_issue_frac"] = fc["SQ_INSTS_VALU"] * 4.0 / 1024.0 / e["kernel_cycles"]
        out[k if cfg == "C2" else k + ":" + cfg] = e
    for kind in ("kt", "kt1"):
        f = glob.glob(os.path.join(BASE, "%s_%s" % (kind, cfg), "**", "*kernel_stats.csv"), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(ROOT, "profiles", "%s_kernel_stats_%s%s.csv" % (tag, cfg, "_one_in_flight" if kind == "kt1" else "")))
# which build the counters were collected on: bench.py compares it with the library it times (roofline.counters_stale)
try:
    out["_build"] = {"librt_hip_so_sha256": open(os.path.join(BASE, "so_sha256.txt")).read().strip()}
except OSError:
    pass
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w"), indent=1)
for k, v in out.items():
    if k.startswith("_"):
        continue
    print("%-28s hbm %8.1f MB/launch  VALU insts %12.0f  valu_issue_frac %s" % (k, v["hbm_bytes_per_launch"] / 1e6, v.get("SQ_INSTS_VALU", 0), v.get("valu_issue_frac")))
