#!/bin/bash
# Round 3, first GPU pass: parity of the split training forward, micro-benchmarks, A/B of the whole step, full GPU suite, kernel trace.
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout 600 python -m pytest tests/test_proj_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "proj or probe or pack32 or attn_fwd or attn_bwd_hm or encoder_autograd" > $O/pytest_new.log 2>&1; echo "new tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_new.log
timeout 300 python tools/kbench_proj.py > $O/kbench_k2.log 2>&1; tail -12 $O/kbench_k2.log
NR_PROJ_KSPLIT=1 KB_ONLY=proj timeout 200 python tools/kbench_proj.py > $O/kbench_k1.log 2>&1; tail -5 $O/kbench_k1.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), json.dumps(d.get('kernel_breakdown', {}))[:1500])"; }
for sp in 1 0 1 0; do
  NR_FWD_SPLIT=$sp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>$O/bench_split$sp.err | tee $O/bench_split$sp.json | line split$sp
done
NR_FWD_SPLIT=1 NR_PROJ_KSPLIT=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-extras 2>/dev/null | line split1_k1
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-extras > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB $O/kernel_stats_NRMS.csv > /dev/null
rm -rf $O/prof
head -30 $O/kernel_stats_NRMS.csv | cut -c1-150
