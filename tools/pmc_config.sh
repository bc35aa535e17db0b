#!/bin/bash
# on the GPU box: the SQ counter set for one config's kernels:  tools/pmc_config.sh C4 tag [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C=$1; TAG=$2; shift; shift
O=gpurun_out/pmc_${TAG}_$C; mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O -o pmc --output-format csv -- python3 bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline "$@" > $O.log 2>&1 || { tail -n 5 $O.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob('$O/**/pmc_counter_collection.csv', recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')[:40]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('GRBM_GUI_ACTIVE',0))[:8]:
    cyc=v['GRBM_GUI_ACTIVE']/8
    print('%-40s launches %4d cycles %.3g valu_issue(4c) %.3f salu/CU %.3f waves %.3g wave_cycles/waves %.3g wait_any %.2f wait_inst %.2f active_valu %.2f' % (k, n[k], cyc, v['SQ_INSTS_VALU']*4/1024/cyc, v['SQ_INSTS_SALU']/256/cyc, v['SQ_WAVES'], v['SQ_WAVE_CYCLES']/max(1,v['SQ_WAVES']), v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES'], v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES'], v['SQ_ACTIVE_INST_VALU']/v['SQ_WAVE_CYCLES']))
PY
