#!/bin/bash
# rocprofv3 evidence for the stand-alone gather probe (north_star: "rocprof HBM GB/s ... for the embedding gather"): kernel durations
# (--kernel-trace --stats) and HBM traffic (FETCH_SIZE / WRITE_SIZE in SEPARATE --pmc passes) at both points.  Usage: bash tools/gather_prof.sh OUTDIR
export TMPDIR=/tmp
O=$1; mkdir -p $O
for P in workload hbm; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gather_${P}_stats -o g -- python tools/gather_probe.py $P > $O/gather_${P}_stats.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gather_${P}_fetch -o g -- python tools/gather_probe.py $P > $O/gather_${P}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/gather_${P}_write -o g -- python tools/gather_probe.py $P > $O/gather_${P}_write.log 2>&1
done
python tools/gather_prof_summary.py $O | tee $O/gather_rocprof.txt
rm -rf $O/gather_*_stats $O/gather_*_fetch $O/gather_*_write
