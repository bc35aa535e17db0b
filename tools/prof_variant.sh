#!/bin/bash
# on the GPU box: rocprofv3 kernel stats of one bench config for a variant library: tools/prof_variant.sh <variant> <config>
V=$1; C=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_var/${V}_$C; mkdir -p $O
RT_HIP_LIB=../../variants/librt_hip_$V.so timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline > $O.log 2>&1 || { tail -n 5 $O.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob('$O/**/kt_kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print('== $V $C')
for r in rows[:8]: print('%-70s calls %5s total %10.3f ms  avg %9.3f ms  %5s%%' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6, r['Percentage']))
PY
