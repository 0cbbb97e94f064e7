#!/bin/bash
# Round-6 profile pass: HBM traffic (FETCH / WRITE in separate --pmc passes) and SQ counters of the dominant kernels, the whole-step traffic
# table, rocprofv3 kernel statistics + gap analysis of the three workloads.  Usage: bash tools/gpu_r06_profiles.sh [TAG]
export TMPDIR=/tmp
TAG=${1:-r06prof}
O=gpurun_out/$TAG
mkdir -p $O
for K in proj_train attn_pool_fwd attn_bwd_hm pool_flat pool_fwd_flat50 pool_flat50_act cgemm_dgrad50 conv_abs; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${K}_fetch -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${K}_write -o pmc -- python tools/prof_kernel.py $K > $O/pmc_${K}_write.log 2>&1
done
python tools/pmc_traffic.py $O profiles/r06_pmc_traffic.txt | tee $O/pmc_traffic.txt
rm -rf $O/pmc_*_fetch $O/pmc_*_write
for K in "attn_bwd_hm attn_bwd" "proj_train qkv_proj" "attn_pool_fwd attn_fwd_kernel" "pool_flat pool3_bwd" "pool_flat50_act pool3_bwd" "dx_gemm dx_gemm" "tn_gemm gemm_ring" "pool_fwd_flat50 pool4_fwd" "cgemm_dgrad50 conv_gemm" "conv_abs conv3_kernel"; do
  set -- $K
  bash tools/pmc_kernel.sh $1 $2 $O/pmc_sq_$1 > /dev/null 2>&1
done
python tools/pmc_sq.py $O r06 | tee $O/pmc_sq_summary.txt
# whole-step traffic: the eager step (same kernels as the graph), 2 warm-up + 2 profiled + 5 timed = 9 identical steps per run
B="--no-graph --steps 5 --warmup 2 --no-parity --no-cpu-baseline --no-extras --no-train-parity"
for M in NRMS NAML; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/step_${M}_small_fetch -o pmc -- python bench.py --model $M $B > $O/step_${M}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/step_${M}_small_write -o pmc -- python bench.py --model $M $B > $O/step_${M}_write.log 2>&1
  python tools/pmc_step_traffic.py $O ${M}_small 9 > $O/step_traffic_${M}.log 2>&1; head -16 $O/step_traffic_${M}.log
  rm -rf $O/step_${M}_small_fetch $O/step_${M}_small_write
done
for W in "NRMS small" "NAML small" "LSTUR large"; do
  set -- $W
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_$1_$2 -o bench -- python bench.py --model $1 --shape $2 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extras --no-train-parity > $O/under_rocprof_$1_$2.log 2>&1
  DB=$(find $O/prof_$1_$2 -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_$1_$2.csv > /dev/null && python tools/rocpd_gaps.py $DB > $O/gaps_$1_$2.txt 2>&1
  rm -rf $O/prof_$1_$2
done
head -16 $O/kernel_stats_NRMS_small.csv | cut -c1-170
