#!/bin/bash
# on the GPU box: C4 and C5 (bench.py --config) for the given variant libraries; stops at the first failure
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/variants
for v in "$@"; do
  for c in C4 C5; do
    RT_HIP_LIB=../../variants/librt_hip_$v.so timeout -k 10 280 python3 -u bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/variants/${v}_$c.json 2> gpurun_out/variants/${v}_$c.err || { echo "$v $c FAILED"; tail -n 5 gpurun_out/variants/${v}_$c.err; exit 1; }
    echo "$v $c: $(cat gpurun_out/variants/${v}_$c.json | cut -c1-600)"
  done
done
