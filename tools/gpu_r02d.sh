#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02d}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
r = d["roofline"]
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 2), "|", r["kernel"], round(r["frac"], 3), round(r["avg_us"], 1))
if "value_dropin" in d: print("   dropin", round(d["value_dropin"]["value"]), round(d["value_dropin"]["ms_per_step"], 2), "score_eval", round(d["score_eval"]["value"]))
print("   kernels", dict(list(d["kernel_breakdown_us_per_step"].items())[:16]))
PY
}
for V in 4 0; do
  NR_ADD_VARIANT=$V timeout 600 python bench.py --no-parity --no-cpu-baseline --no-extras > $O/bench_NRMS_add$V.json 2> $O/bench_NRMS_add$V.err; q $O/bench_NRMS_add$V.json
done
NR_ADD_VARIANT=0 timeout 600 python bench.py --model LSTUR --no-parity --no-cpu-baseline --no-extras > $O/bench_LSTUR_add0.json 2> $O/bench_LSTUR_add0.err; q $O/bench_LSTUR_add0.json
timeout 600 python bench.py --model LSTUR --no-parity --no-cpu-baseline > $O/bench_LSTUR.json 2> $O/bench_LSTUR.err; q $O/bench_LSTUR.json
timeout 600 python bench.py --model NAML --no-parity --no-cpu-baseline > $O/bench_NAML.json 2> $O/bench_NAML.err; q $O/bench_NAML.json
timeout 600 python bench.py --no-parity --no-cpu-baseline > $O/bench_NRMS.json 2> $O/bench_NRMS.err; q $O/bench_NRMS.json
