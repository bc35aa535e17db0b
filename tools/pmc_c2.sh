#!/bin/bash
# on the GPU box: two counter passes of the C2 frame kernel (one frame in flight), summed per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_c2; mkdir -p $O
B="bench.py --steps 2 --warmup 1 --frames-per-step 8 --no-cpu-baseline --no-extras --streams 1"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES -d $O/a -o pmc --output-format csv -- python3 $B > $O/a.log 2>&1 || { tail -n 5 $O/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $O/b -o pmc --output-format csv -- python3 $B > $O/b.log 2>&1 || { tail -n 5 $O/b.log; exit 1; }
python3 - <<PY
import csv,glob,collections
for tag in 'ab':
    f=glob.glob('$O/%s/**/pmc_counter_collection.csv'%tag, recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'frame_direct' in k: print(k, {a: '%.4g' % (sum(b)/len(b)) for a,b in v.items()})
PY
