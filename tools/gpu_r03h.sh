#!/bin/bash
# Round 3, pass h: hand-written weight-gradient GEMM (transposing LDS reads) vs chunked hipBLASLt: parity, kernel and step A/B, LDS bank-conflict counters.
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
timeout 900 python -m pytest tests/test_proj_gpu.py -x -q -m gpu -k "tn_gemm or dx_gemm or encoder_autograd" > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed|Error" $O/pytest_new.log | cut -c1-600 | tail -6
KB_ONLY=dW timeout 300 python tools/kbench_proj.py > $O/kbench_dw.log 2>&1; tail -8 $O/kbench_dw.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d['kernel_breakdown_us_per_step']; print('$1', round(d['value']), round(d['ms_per_step'],3), {k: v for k, v in kb.items() if 'dW' in k or 'tn_gemm' in k or 'unpack' in k})"; }
for v in "NR_WGRAD_GEMM=1" "NR_WGRAD_GEMM=0" "NR_WGRAD_GEMM=1" "NR_WGRAD_GEMM=0"; do
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench.err | tee "$O/bench_$(echo $v | tr ' =' '__').json" | line "$v"
done
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_optim_gpu.py -x -q -m gpu > $O/pytest_model.log 2>&1; echo "model+optim tests rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest_model.log
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_tn -o pmc -- env KB_ONLY=dWqkv_tn python tools/kbench_proj.py > $O/pmc_tn.log 2>&1
python tools/pmc_summary.py $O/pmc_tn tn_gemm | tee $O/pmc_sq_tn_gemm.txt
python tools/pmc_summary.py $O/pmc_tn dx_gemm | tee $O/pmc_sq_dx_gemm.txt
rm -rf $O/pmc_tn
