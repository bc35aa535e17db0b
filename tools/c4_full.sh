#!/bin/bash
# on the GPU box: C4 at the full 256 spp, in-place and deferred Newton solves
cd "$GRAFT_REPO_ROOT"
for D in 0 1; do
  RT_HIP_PARAM_DEFER=$D timeout -k 10 250 python bench.py --config C4 --spp 256 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/c4_full_$D.json 2> gpurun_out/c4_full_$D.err || { tail -n 3 gpurun_out/c4_full_$D.err; exit 1; }
  python - $D <<PY
import json,sys
d=json.load(open('gpurun_out/c4_full_%s.json' % sys.argv[1]))
print('DEFER', sys.argv[1], 'ms_per_step', d['ms_per_step'], 'Mrays/s', d['value'])
PY
done
