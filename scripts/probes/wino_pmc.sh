#!/bin/bash
# usage on the GPU box: scripts/probes/wino_pmc.sh  -> counters of the Winograd prototype kernel
cd $GRAFT_REPO_ROOT/scripts/probes && hipcc --offload-arch=gfx950 -O3 -o /tmp/wp wino_proto.hip > /tmp/cc.log 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/wino_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_UNALIGNED_STALL SQ_WAVES"
P3="SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM"
P4="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"
P5="FETCH_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P -f csv -d $OUT/p$i -o p$i -- /tmp/wp time "$@" > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for i in range(1,6):
    for f in glob.glob("$OUT/p%d/*counter_collection.csv"%i):
        agg=collections.defaultdict(lambda:[0,0])
        for r in csv.DictReader(open(f)):
            if 'wino' not in r['Kernel_Name']: continue
            a=agg[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
        for k,(v,n) in agg.items(): print("p%d %-34s %16.0f per launch"%(i,k,v/n))
PY
