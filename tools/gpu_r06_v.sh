#!/bin/bash
# twenty-second GPU pass of round 6: (1) the flat pooling backward with a run-time row stride of the sequence gradient against a build with the
# constant stride (tools/ab/libnr_engine_old.so), stand-alone, one box; (2) LSTUR with the title third of the gradient read in place vs through
# a contiguous copy; (3) the N > 1 code path of bench.py on the new step (two ranks on this one GPU, gloo)
export TMPDIR=/tmp
O=gpurun_out/r06v
mkdir -p $O
pk() { for K in pool_flat pool_flat_act pool_flat50_act; do for r in 1 2 3; do timeout 120 python tools/prof_kernel.py $K 2>/dev/null | tail -1; done; done; }
echo "=== NEW (run-time stride)" | tee $O/ab_pool3_gstride.txt; pk | tee -a $O/ab_pool3_gstride.txt
for r in 1 2 3; do timeout 120 python tools/prof_kernel.py pool_flat_act_gs 2>/dev/null | tail -1; done | tee -a $O/ab_pool3_gstride.txt
cp news_recommendation_amd/libnr_engine.so /tmp/libnr_engine_new.so
cp tools/ab/libnr_engine_old.so news_recommendation_amd/libnr_engine.so
echo "=== OLD (constant stride)" | tee -a $O/ab_pool3_gstride.txt; pk | tee -a $O/ab_pool3_gstride.txt
cp /tmp/libnr_engine_new.so news_recommendation_amd/libnr_engine.so
echo "=== NEW again" | tee -a $O/ab_pool3_gstride.txt; pk | tee -a $O/ab_pool3_gstride.txt
for F in 1 0 1 0; do
NR_POOL_G_STRIDED=$F timeout 600 python bench.py --model LSTUR --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/err.txt | grep '^{' | tail -1 > $O/line_LSTUR_$F.json
python - <<PY | tee -a $O/ab_pool3_gstride.txt
import json
d = json.load(open("$O/line_LSTUR_$F.json")); kb = d["kernel_breakdown_us_per_step"]
print("LSTUR strided=$F ms", round(d["ms_per_step"], 3), {k: v for k, v in kb.items() if "additive_bwd" in k})
PY
done
bash tools/gpu_two_ranks_one_gpu.sh 2>&1 | tail -30
cp gpurun_out/two_ranks_*.log $O/ 2>/dev/null
