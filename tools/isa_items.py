#!/usr/bin/env python3
"""Static side of the instruction itemisation (profiles/README.md): the loops of one kernel in a `--save-temps` .s file with the instructions
each holds outside its child loops, by class -- VALU (of it FP64 / FP32 / moves+selects+compares / cross-lane / integer), SALU, branch,
scalar memory, LDS, vector memory.  The dynamic side (how often each loop body runs per wave) comes from a -DRT_TIMERS build
(tools/timer_probe.py); `--trips file.json` multiplies the two: {"<loop label>": [name, trips_per_wave], "top": [name, 1]}.

  hipcc ... --save-temps ; tools/isa_items.py rt_hip-hip-amdgcn-amd-amdhsa-gfx950.s k_frame_directILb0ELb0ELb1EE [--trips t.json]
"""
import collections
import json
import re
import sys

CLASSES = ["valu", "v_f64", "v_f32", "v_int", "v_mov_sel_cmp", "v_lane", "v_cvt", "salu", "branch", "smem", "lds", "vmem", "wait_nop"]


def classify(op):
    c = collections.Counter()
    if op.startswith("v_"):
        c["valu"] = 1
        if re.search(r"_f64|_f64_", op) and not op.startswith("v_cvt") and not op.startswith("v_cmp") and not op.startswith("v_cmpx") and "cndmask" not in op:
            c["v_f64"] = 1
        elif op.startswith("v_cvt"):
            c["v_cvt"] = 1
        elif op.startswith(("v_mov", "v_cndmask", "v_cmp", "v_cmpx", "v_accvgpr")):
            c["v_mov_sel_cmp"] = 1
        elif op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "v_permlane", "v_bpermute")) or "dpp" in op:
            c["v_lane"] = 1
        elif re.search(r"_f32|_f16", op):
            c["v_f32"] = 1
        else:
            c["v_int"] = 1
    elif op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        c["smem"] = 1
    elif op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_call")):
        c["branch"] = 1
    elif op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
        c["wait_nop"] = 1
    elif op.startswith("s_"):
        c["salu"] = 1
    elif op.startswith("ds_"):
        c["lds"] = 1
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        c["vmem"] = 1
    return c


def main():
    path, pat = sys.argv[1], sys.argv[2]
    trips = None
    if "--trips" in sys.argv:
        trips = json.load(open(sys.argv[sys.argv.index("--trips") + 1]))
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith("\t.section"))
    blocks = []          # (label, innermost loop header or None, depth, counter)
    cur = {"label": "entry", "loop": None, "depth": 0, "cnt": collections.Counter(), "hdr": False}
    blocks.append(cur)
    parents = {}
    i = start + 1
    while i < end:
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", l)
        if m:
            label = m.group(1) or ("bb.%s" % m.group(2))
            # the block's comment lines: this one and the pure-comment lines that follow
            com = [l]
            j = i + 1
            while j < end and re.match(r"^\s+;", lines[j]):
                com.append(lines[j]); j += 1
            text = "\n".join(com)
            loop, depth, hdr = None, 0, False
            mh = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", text)
            mi = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", text)
            if mh:
                loop, depth, hdr = label.replace(".L", ""), int(mh.group(1)), True
                par = re.findall(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", text)
                parents[loop] = par[-1][0] if par else None
            elif mi:
                loop, depth = mi.group(1), int(mi.group(2))
            cur = {"label": label, "loop": loop, "depth": depth, "cnt": collections.Counter(), "hdr": hdr}
            blocks.append(cur)
            i = j
            continue
        mo = re.match(r"^\s+([a-z_0-9]+)", l)
        if mo and not l.strip().startswith((";", ".")):
            cur["cnt"] += classify(mo.group(1))
        i += 1
    # top-level code is split at every depth-1 loop it lies between, so that setup / confirmation / shading can be told apart
    # --blocks LOOP[,LOOP...]: the basic blocks of these loops are listed one by one (a loop nest the compiler has folded into one loop --
    # the cull step with its hand-back rounds -- is told apart by block)
    by_block = set()
    if "--blocks" in sys.argv:
        by_block = set(sys.argv[sys.argv.index("--blocks") + 1].split(","))
    regions = collections.OrderedDict()
    seg = 0
    last_top_loop = None
    for b in blocks:
        if b["loop"] is None:
            key = "top#%d" % seg
        else:
            key = b["loop"]
            top = key
            while parents.get(top):
                top = parents[top]
            if top != last_top_loop:
                last_top_loop = top; seg += 1
            if b["loop"] in by_block and not b["hdr"]:
                key = b["loop"] + ":" + b["label"].replace(".L", "")
                parents[key] = b["loop"]
        regions.setdefault(key, collections.Counter())
        regions[key] += b["cnt"]
    def depth_of(k):
        d, x = 0, k
        while parents.get(x):
            d += 1; x = parents[x]
        return d + 1 if not k.startswith("top#") else 0
    print("%-28s %s" % ("region (loop header)", " ".join("%7s" % c[:7] for c in CLASSES)))
    total = collections.Counter()
    weighted = collections.Counter()
    for k, c in regions.items():
        name = ("  " * depth_of(k)) + k
        t = None
        if trips and k in trips:
            name += " " + trips[k][0]; t = float(trips[k][1])
        print("%-28s %s%s" % (name[:28], " ".join("%7d" % c[x] for x in CLASSES), ("   x %.2f" % t) if t is not None else ""))
        total += c
        if t is not None:
            for x in CLASSES:
                weighted[x] += c[x] * t
    print("%-28s %s" % ("static total", " ".join("%7d" % total[x] for x in CLASSES)))
    if trips:
        print("%-28s %s" % ("dynamic per wave", " ".join("%7d" % weighted[x] for x in CLASSES)))
        if "--by-name" in sys.argv:
            byname = collections.OrderedDict()
            for k, c in regions.items():
                if k in trips:
                    n, t = trips[k][0], float(trips[k][1])
                    byname.setdefault(n, collections.Counter())
                    for x in CLASSES:
                        byname[n][x] += c[x] * t
            print()
            print("%-28s %s" % ("item (per wave)", " ".join("%7s" % c[:7] for c in CLASSES)))
            for n, c in byname.items():
                print("%-28s %s" % (n[:28], " ".join("%7d" % c[x] for x in CLASSES)))


if __name__ == "__main__":
    main()
