#!/bin/bash
cd "$GRAFT_REPO_ROOT"
run() { L=$1; C=$2; shift; shift
  X=""; [ $C = C4 ] && X="--spp 32"
  env "$@" timeout -k 10 200 python bench.py --config $C --steps 1 --warmup 1 --no-cpu-baseline $X > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { tail -n 3 gpurun_out/sweep.err; return 1; }
  python - "$L" $C <<PY
import json,sys
d=json.load(open('gpurun_out/sweep.json')); c=d['config']
print(sys.argv[1], sys.argv[2], 'ms_per_step', d['ms_per_step'], 'eye', c.get('eye_pass_s'))
PY
}
run default C5 X=1 || exit 1
run w2048 C5 RT_HIP_SOLVE_WAVES_ROT=2048 || exit 1
run w1024 C5 RT_HIP_SOLVE_WAVES_ROT=1024 || exit 1
run batch4_w2048 C5 RT_HIP_SOLVE_BATCH=4 RT_HIP_SOLVE_WAVES_ROT=2048 || exit 1
run default C4 X=1 || exit 1
run w1024 C4 RT_HIP_SOLVE_WAVES_BEZ=1024 || exit 1
