#!/bin/bash
# Runs on the GPU box (gpurun): kernel trace of the default bench command, then separate PMC passes (one frame in flight:
# counters are per dispatch and must not overlap).  Output under gpurun_out/prof_*; tools/make_traffic_json.py digests it.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt --output-format csv -- python3 $B > gpurun_out/prof_kt.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt1 -o kt --output-format csv -- python3 $B --streams 1 > gpurun_out/prof_kt1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d gpurun_out/prof_pmc_a -o pmc --output-format csv -- python3 $B --streams 1 > gpurun_out/prof_pmc_a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d gpurun_out/prof_pmc_b -o pmc --output-format csv -- python3 $B --streams 1 > gpurun_out/prof_pmc_b.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_pmc_c -o pmc --output-format csv -- python3 $B --streams 1 > gpurun_out/prof_pmc_c.log 2>&1
tail -1 gpurun_out/prof_kt.log
ls gpurun_out/prof_kt gpurun_out/prof_pmc_a
