#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel traces and separate PMC passes of bench.py, for the default C2 line and for
# --config C3 / C4 / C5.  One frame in flight for the counters (they are per dispatch and must not overlap).  Output under
# gpurun_out/prof_<tag>_*; tools/make_traffic_json.py digests it into profiles/<tag>_traffic.json.
#   tools/collect_profiles.sh r02 [C2 C3 C4 C5]
set -e
TAG=${1:-r03}; shift || true
CONFIGS=${@:-C2 C3 C4 C5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG
mkdir -p $O
for C in $CONFIGS; do
  if [ $C = C2 ]; then B="bench.py --steps 3 --warmup 1 --frames-per-step 8 --no-cpu-baseline --no-extras"; else B="bench.py --config $C --steps 1 --warmup 0 --no-cpu-baseline"; fi
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt_$C -o kt --output-format csv -- python3 $B > $O/kt_$C.log 2>&1
  [ $C = C2 ] && timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt1_$C -o kt --output-format csv -- python3 $B --streams 1 > $O/kt1_$C.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum -d $O/pmc_a_$C -o pmc --output-format csv -- python3 $B --streams 1 > $O/pmc_a_$C.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -d $O/pmc_b_$C -o pmc --output-format csv -- python3 $B --streams 1 > $O/pmc_b_$C.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_c_$C -o pmc --output-format csv -- python3 $B --streams 1 > $O/pmc_c_$C.log 2>&1
  # the per-type VALU counters (the mix weights bench.py's VALU peak: FP64 4 cycles per wave64 instruction, the rest 2) and the instruction classes
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/pmc_d_$C -o pmc --output-format csv -- python3 $B --streams 1 > $O/pmc_d_$C.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_e_$C -o pmc --output-format csv -- python3 $B --streams 1 > $O/pmc_e_$C.log 2>&1
  tail -c 300 $O/kt_$C.log; echo
done
sha256sum ray_tracing_amd/lib/librt_hip.so | cut -d" " -f1 > $O/so_sha256.txt     # which build the counters belong to (bench.py: roofline.counters_stale)
find $O -name "*.csv" | head -60
