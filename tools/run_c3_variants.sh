#!/bin/bash
# on the GPU box: C3 (bench.py --config C3) for the given variant libraries
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/variants
for v in "$@"; do
  RT_HIP_LIB=../../variants/librt_hip_$v.so timeout -k 10 280 python3 -u bench.py --config C3 --spp 16 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/variants/${v}_C3.json 2> gpurun_out/variants/${v}_C3.err || { echo "$v C3 FAILED"; tail -n 5 gpurun_out/variants/${v}_C3.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/variants/${v}_C3.json')); print('%-10s C3 %8.1f Mrays/s  %.4f s/step' % ('$v', d['value'], d['config']['s_per_step']))"
done
