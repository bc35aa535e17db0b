#!/bin/bash
# on the GPU box: C2 default line (no extras) with the in-tree library and with variants/librt_hip_<name>.so, alternating, same box
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
  for L in base "$@"; do
    if [ $L = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=../../variants/librt_hip_$L.so; fi
    timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -n 3 gpurun_out/ab.err; exit 1; }
    python - $L <<PY
import json,sys
d=json.load(open('gpurun_out/ab.json'))
print(sys.argv[1], 'Mrays/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'])
PY
  done
done
