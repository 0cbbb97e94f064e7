#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02i}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "pool2" > $O/pytest_pool2.log 2>&1; tail -3 $O/pytest_pool2.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
k = d["kernel_breakdown_us_per_step"]
print("| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 2), {a: round(b) for a, b in k.items() if 'additive' in a and '20' in a or 'title' in a and 'additive' in a})
PY
}
for rep in 1 2; do
for cfg in "NR_POOL2_GEOM=44 NR_POOL2_FWD=0" "NR_POOL2_GEOM=28 NR_POOL2_FWD=0" "NR_POOL2_GEOM=28 NR_POOL2_FWD=1" "NR_POOL2_GEOM=44 NR_POOL2_FWD=1"; do
  env $cfg timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b.json 2> $O/b.err; echo -n "$cfg "; q $O/b.json
done
done
