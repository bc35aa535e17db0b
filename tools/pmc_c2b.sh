#!/bin/bash
# on the GPU box: instruction-cache and scalar-data-cache counters of the C2 frame kernel (one frame in flight)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_c2b; mkdir -p $O
B="bench.py --steps 2 --warmup 1 --frames-per-step 8 --no-cpu-baseline --no-extras --streams 1"
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY -d $O/a -o pmc --output-format csv -- python3 $B > $O/a.log 2>&1 || { tail -n 8 $O/a.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob('$O/a/**/pmc_counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'frame_direct' in k: print(k, {a: '%.4g' % (sum(b)/len(b)) for a,b in v.items()})
PY
