#!/bin/bash
# r02n: pooling backward with the activation gradient fused into its epilogue (nr_additive_bwd_act): parity + NAML / LSTUR A/B
export TMPDIR=/tmp
O=gpurun_out/${1:-r02n}
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "additive or conv or naml or lstur or knobs" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
k = d["kernel_breakdown_us_per_step"]
print("| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 2), {a: round(b) for a, b in k.items() if 'additive_bwd' in a or 'act_bwd' in a}, 'sum', round(sum(k.values())))
PY
}
for rep in 1 2; do
for cfg in "NR_POOL_ACT_FUSE=0" "NR_POOL_ACT_FUSE=1"; do
  env $cfg timeout 300 python bench.py --model NAML --no-parity --no-cpu-baseline --no-extras --steps 30 > $O/b.json 2> $O/b.err; echo -n "NAML $cfg "; q $O/b.json
  env $cfg timeout 300 python bench.py --model LSTUR --shape large --no-parity --no-cpu-baseline --no-extras --steps 30 > $O/b.json 2> $O/b.err; echo -n "LSTUR $cfg "; q $O/b.json
done
done
