#!/bin/bash
# round 4, call q: flat pooling backward, g_out pieces requested up front -- phases, timeline, flat-only step times (baselines: call p of the same day)
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r04q}; mkdir -p $O
timeout 300 python tools/pool3_phases.py --timeline > $O/pool3_phases.txt 2> $O/pool3_phases.err; grep -v "^  wg" $O/pool3_phases.txt
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
kb = d["kernel_breakdown_us_per_step"]
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "loss", round(d["loss"], 4), {k: v for k, v in kb.items() if 'additive_bwd' in k})
PY
}
B="--no-parity --no-cpu-baseline --no-extras --no-train-parity"
run() { tag=$1; m=$2; shift 2; env "$@" timeout 600 python bench.py --model $m $B > $O/$tag.json 2> $O/$tag.err; q $O/$tag.json; }
for m in NRMS NAML LSTUR; do run ${m}_flat1_a $m NR_POOL_FLAT=1; run ${m}_flat0_a $m NR_POOL_FLAT=0; done
