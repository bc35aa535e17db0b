#!/bin/bash
# on the GPU box: the per-type VALU instruction counters of the C2 frame kernel (one frame in flight), two passes of 8 SQ counters each,
# averaged per kernel.  Feeds the mix-weighted VALU peak of bench.py (FP64 4 cycles per wave64 instruction, everything else 2).
#   tools/pmc_types.sh [tag] [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-types}; shift || true
O=gpurun_out/pmc_$TAG; mkdir -p $O
B="bench.py --steps 2 --warmup 1 --frames-per-step 8 --no-cpu-baseline --no-extras --streams 1 $@"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/a -o pmc --output-format csv -- python3 $B > $O/a.log 2>&1 || { tail -n 5 $O/a.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH -d $O/b -o pmc --output-format csv -- python3 $B > $O/b.log 2>&1 || { tail -n 5 $O/b.log; exit 1; }
python3 - <<PY
import csv,glob,collections,json
out={}
for tag in 'ab':
    f=glob.glob('$O/%s/**/pmc_counter_collection.csv'%tag, recursive=True)[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        if 'k_' in k and '<true' not in k:
            out.setdefault(k,{}).update({a: sum(b)/len(b) for a,b in v.items()})
            out[k]['launches']=len(next(iter(v.values())))
json.dump(out, open('$O/types.json','w'), indent=1)
for k,v in out.items():
    if 'frame_direct' in k or 'shadow' in k: print(k, {a: '%.4g' % b for a,b in v.items()})
PY
