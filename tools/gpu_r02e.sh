#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r02e}
mkdir -p $O
q() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("NO JSON", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:]); sys.exit(0)
print(sys.argv[1].split('/')[-1], "| value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "host", round(d["host_enqueue_ms_per_step"], 2), "sumk", round(sum(d["kernel_breakdown_us_per_step"].values())))
PY
}
timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b_default.json 2> $O/b_default.err; q $O/b_default.json
OMP_NUM_THREADS=16 timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b_omp16.json 2> $O/b_omp16.err; q $O/b_omp16.json
timeout 300 python bench.py --no-parity --no-cpu-baseline --no-extras --steps 40 > $O/b_default2.json 2> $O/b_default2.err; q $O/b_default2.json
timeout 400 rocprofv3 --kernel-trace -d $O/prof -o bench -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-parity --no-extras > $O/under_rocprof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_gaps.py $DB | tee $O/gaps.txt
rm -rf $O/prof
