#!/bin/bash
# on the GPU box: rocprofv3 kernel statistics of bench.py --config <Cn> [extra bench args]; prints the top kernels.
#   tools/prof_config.sh C4 tag [--spp 16]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C=$1; TAG=$2; shift; shift
O=gpurun_out/kt_${TAG}_$C; mkdir -p $O
timeout -s KILL 280 rocprofv3 --kernel-trace --stats -d $O -o kt --output-format csv -- python3 bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O.log 2>&1 || { tail -n 5 $O.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob('$O/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-70s calls %6s avg %10.1f us total %9.1f ms %6s %%' % (r['Name'].split('(')[0].replace('void ','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
tail -c 400 $O.log
