#!/bin/bash
# r02k: persistent GRU kernels (parity, A/B), nr_wgrad_unpack (parity through the optimiser lock-step tests, step time)
export TMPDIR=/tmp
O=gpurun_out/${1:-r02k}
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "gru or wgrad or lstur or optim" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
k = d["kernel_breakdown_us_per_step"]
print("| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 2), {a: round(b) for a, b in k.items() if 'gru' in a or 'unpack' in a}, 'sum', round(sum(k.values())))
PY
}
for rep in 1 2; do
for cfg in "NR_GRU_PERSIST=0" "NR_GRU_PERSIST=1"; do
  env $cfg timeout 300 python bench.py --model LSTUR --shape large --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err; echo -n "LSTUR $cfg "; q $O/b.json
done
done
for rep in 1 2; do
for cfg in "NR_WGRAD_UNPACK=0" "NR_WGRAD_UNPACK=1"; do
  env $cfg timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err; echo -n "NRMS $cfg "; q $O/b.json
done
done
