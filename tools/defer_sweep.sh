#!/bin/bash
# on the GPU box: C4 (--spp 32) and C5 step times for a few settings of the deferred solver's knobs
cd "$GRAFT_REPO_ROOT"
run() { # label, env...
  L=$1; shift
  for C in C4 C5; do
    X=""; [ $C = C4 ] && X="--spp 32"
    env "$@" timeout -k 10 200 python bench.py --config $C --steps 1 --warmup 1 --no-cpu-baseline $X > gpurun_out/sweep.json 2> gpurun_out/sweep.err || { tail -n 3 gpurun_out/sweep.err; return 1; }
    python - "$L" $C <<PY
import json,sys
d=json.load(open('gpurun_out/sweep.json')); c=d['config']
print(sys.argv[1], sys.argv[2], 'ms_per_step', d['ms_per_step'], 'eye', c.get('eye_pass_s'), 'parity', json.dumps(d.get('parity'))[:160] if d.get('parity') else None)
PY
  done
}
run inplace RT_HIP_PARAM_DEFER=0 || exit 1
run default X=1 || exit 1
run waves_bez4096 RT_HIP_SOLVE_WAVES_BEZ=4096 || exit 1
run waves_bez1024_rot2048 RT_HIP_SOLVE_WAVES_BEZ=1024 RT_HIP_SOLVE_WAVES_ROT=2048 || exit 1
run batch2 RT_HIP_SOLVE_BATCH=2 || exit 1
run aa1 RT_HIP_AA_BATCH_GIB=1 || exit 1
