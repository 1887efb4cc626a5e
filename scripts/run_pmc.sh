#!/bin/bash
# usage (on the GPU box, from repo root): scripts/run_pmc.sh <tag> [bench args]
# separate --pmc passes (SQ 8 slots, TCC: FETCH_SIZE and WRITE_SIZE cannot share a pass)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_UNALIGNED_STALL SQ_WAVES"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"   # L2 / vector-L1 request counts
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --soak-seconds 0 --no-cpu-baseline --no-secondary --no-inference "$@" > $OUT/p$i.log 2>&1
  rm -f $OUT/p$i/*kernel_trace.csv $OUT/p$i/*.db
done
KSTATS=${KSTATS:-} python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT conv_wino64_kernel conv_up3_kernel conv_wino_kernel conv_wino_wgrad_kernel conv_fwd_kernel conv_wgrad_kernel conv_wgrad3_kernel conv_wgrad3s_kernel conv_thin_n thin1_ bn_bwd pool27 pool_fwd head_fused_kernel head_dgrad_kernel reduce_splits adam > $OUT/summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +8M -delete
head -60 $OUT/summary.txt
