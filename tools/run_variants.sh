#!/bin/bash
# on the GPU box: bench the given variant libraries (C2, no extras); prints value / ms per frame / one-in-flight ms / kernel ms and the parity object
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/variants
for v in "$@"; do
  RT_HIP_LIB=../../variants/librt_hip_$v.so timeout -k 10 200 python3 -u bench.py --steps 6 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/variants/$v.json 2> gpurun_out/variants/$v.err || { echo "$v FAILED"; tail -n 3 gpurun_out/variants/$v.err; exit 1; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/variants/$v.json')); print('%-14s value %8.1f  ms/frame %.4f  one-in-flight %.4f  kernel %.4f  parity %s' % ('$v', d['value'], d['config']['ms_per_frame'], d['config']['ms_per_frame_one_in_flight'], d['roofline']['kernel_ms'], json.dumps(d.get('parity'))[:300]))"
done
