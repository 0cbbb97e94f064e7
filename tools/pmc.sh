#!/bin/bash
# Collect PMC counters for one engine kernel (separate --pmc passes, no trace domains).  Usage: tools/pmc.sh KERNEL_TAG OUTDIR
export TMPDIR=/tmp
K=$1; OUT=$2; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="WRITE_SIZE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o pmc -- python tools/prof_kernel.py $K > $OUT/p$i.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head
