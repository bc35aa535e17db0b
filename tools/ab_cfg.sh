#!/bin/bash
# on the GPU box: one config's step time with the in-tree library and with variants/librt_hip_<name>.so:  tools/ab_cfg.sh C3 name [bench args]
cd "$GRAFT_REPO_ROOT"
C=$1; V=$2; shift; shift
for rep in 1 2; do for L in base $V; do
  if [ $L = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=../../variants/librt_hip_$L.so; fi
  timeout -k 10 250 python bench.py --config $C --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/abc.json 2> gpurun_out/abc.err || { tail -n 3 gpurun_out/abc.err; exit 1; }
  python - $L $C <<PY
import json,sys
d=json.load(open('gpurun_out/abc.json')); print(sys.argv[2], sys.argv[1], 'ms_per_step', d['ms_per_step'], 'eye', d['config'].get('eye_pass_s'), 'parity', json.dumps(d.get('parity'))[-200:] if d.get('parity') else None)
PY
done; done
