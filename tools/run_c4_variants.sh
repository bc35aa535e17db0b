#!/bin/bash
# on the GPU box: C4 only for the given variants
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/variants
for v in "$@"; do
  RT_HIP_LIB=../../variants/librt_hip_$v.so timeout -k 10 280 python3 -u bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/variants/${v}_C4.json 2> gpurun_out/variants/${v}_C4.err || { echo "$v C4 FAILED"; tail -n 5 gpurun_out/variants/${v}_C4.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/variants/${v}_C4.json')); print('%-10s C4 %8.1f Mrays/s  %.4f s/step' % ('$v', d['value'], d['config']['s_per_step']))"
done
