#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02f}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
k = d["kernel_breakdown_us_per_step"]
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 2), "sumk", round(sum(k.values())),
      {a: round(b) for a, b in k.items() if 'additive' in a or 'gemm_dW' in a or 'attn_bwd[S=20]' in a or 'conv3_dgrad' in a})
PY
}
for rep in 1 2; do
for cfg in "NR_WGRAD_OVERLAP=1" "NR_WGRAD_OVERLAP=0" "NR_POOL2_FWD=0"; do
  env $cfg timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err; echo -n "$cfg "; q $O/b.json
done
done
for cfg in "NR_WGRAD_OVERLAP=1" "NR_WGRAD_OVERLAP=0"; do
  env $cfg timeout 300 python bench.py --model LSTUR --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err; echo -n "LSTUR $cfg "; q $O/b.json
  env $cfg timeout 300 python bench.py --model NAML --no-parity --no-cpu-baseline --no-extras --steps 30 > $O/b.json 2> $O/b.err; echo -n "NAML $cfg "; q $O/b.json
done
