#!/bin/bash
# on the GPU box: the given variant libraries on a larger heightfield (CELLS per side, default 1500 = 4.5 M triangles, leaf stream 576 MB)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/variants
CELLS=${CELLS:-1500}
for v in "$@"; do
  RT_HIP_LIB=../../variants/librt_hip_$v.so timeout -k 10 250 python3 -u bench.py --cells $CELLS --steps 3 --warmup 1 --frames-per-step 32 --no-extras --no-cpu-baseline > gpurun_out/variants/big_$v.json 2> gpurun_out/variants/big_$v.err || { echo "$v FAILED"; tail -n 3 gpurun_out/variants/big_$v.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/variants/big_$v.json')); print('%-10s value %8.1f  ms/frame %.4f  triangles %d' % ('$v', d['value'], d['config']['ms_per_frame'], d['config']['triangles']))"
done
