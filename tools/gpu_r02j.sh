#!/bin/bash
# r02j: attn_bwd XCD-major A/B (time + HBM traffic), two ranks on one GPU through bench.py's N > 1 path, LSTUR drop-in host profile.
export TMPDIR=/tmp
O=gpurun_out/${1:-r02j}
mkdir -p $O
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
k = d["kernel_breakdown_us_per_step"]
print("| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 2), {a: round(b) for a, b in k.items() if 'attn_bwd' in a})
PY
}
for rep in 1 2; do
for cfg in "NR_ATTN_XCD=0" "NR_ATTN_XCD=1"; do
  env $cfg timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err; echo -n "$cfg "; q $O/b.json
done
done
for X in 0 1; do
  NR_ATTN_XCD=$X timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_attn_bwd_fetch -o pmc -- python tools/prof_kernel.py attn_bwd > $O/pmc_fetch.log 2>&1
  NR_ATTN_XCD=$X timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_attn_bwd_write -o pmc -- python tools/prof_kernel.py attn_bwd > $O/pmc_write.log 2>&1
  echo "NR_ATTN_XCD=$X"; python tools/pmc_traffic.py $O | tee $O/pmc_traffic_xcd$X.txt
  rm -rf $O/pmc_attn_bwd_fetch $O/pmc_attn_bwd_write
done
timeout 300 python tools/diag_dropin2.py LSTUR large > $O/diag_dropin_LSTUR_large.log 2>&1; head -30 $O/diag_dropin_LSTUR_large.log
timeout 300 python tools/diag_dropin2.py LSTUR small > $O/diag_dropin_LSTUR_small.log 2>&1; head -4 $O/diag_dropin_LSTUR_small.log
bash tools/gpu_two_ranks_one_gpu.sh
timeout 600 python -m pytest tests -m gpu -q -x -k "attn_bwd or lstur or model" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
