#!/bin/bash
# SQ / LDS / memory counters of one engine kernel (separate --pmc passes, no trace domains).  Usage: tools/pmc_kernel.sh PROF_KERNEL_TAG KERNEL_SUBSTR OUTDIR
export TMPDIR=/tmp
K=$1; SUB=$2; OUT=$3; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"
P3="SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o pmc -- python tools/prof_kernel.py $K > $OUT/p$i.log 2>&1
done
python tools/pmc_summary.py $OUT $SUB > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
