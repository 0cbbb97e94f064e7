"""Summarise tools/gather_prof.sh: per point the rocprofv3 average duration of gather_rows_kernel, FETCH_SIZE / WRITE_SIZE per launch (raw KiB;
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH doubled per MI355X_MICROARCH.md's gfx950 calibration), algorithmic read rate vs 8 TB/s.
Writes OUT/gather_traffic.json (what bench.py's gather_roofline.traffic reports, keyed on the kernel-source hash)."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
out = sys.argv[1]
SUB = 'gather_rows_kernel'
TOK = 512 * 53 * 20


def rows(d, pat):
    for f in glob.glob(f'{d}/**/*{pat}*.csv', recursive=True):
        yield from csv.DictReader(open(f))


res = {}
for p in ('workload', 'hbm'):
    durs = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in rows(f'{out}/gather_{p}_stats', 'kernel_trace') if SUB in r['Kernel_Name']]
    cnt = {}
    for c in ('fetch', 'write'):
        v = [float(r['Counter_Value']) for r in rows(f'{out}/gather_{p}_{c}', 'counter_collection') if SUB in r['Kernel_Name'] and r['Counter_Name'] == f'{c.upper()}_SIZE']
        cnt[c] = sum(v) / len(v) if v else None
    if not durs:
        print(p, 'no kernel rows'); continue
    durs = sorted(durs)[:max(1, len(durs) - 2)]          # drop the two slowest (first-touch) launches
    us = sum(durs) / len(durs) / 1e3
    alg = TOK * 1208
    e = {"avg_us_rocprof": us, "launches": len(durs), "algorithmic_read_bytes": alg, "achieved_GBs": alg / us / 1e3, "frac_of_8TBs": alg / us / 1e3 / 8000.0,
         "fetch_kib_raw": cnt['fetch'], "write_kib_raw": cnt['write'], "source_hash": bench.gather_source_hash()}
    if cnt['fetch'] is not None and cnt['write'] is not None:
        e["traffic_bytes"] = int((2 * cnt['fetch'] + cnt['write']) * 1024)
        e["hbm_GBs_moved"] = e["traffic_bytes"] / us / 1e3
    res[p] = e
    print(f"{p}: {us:.1f} us avg over {len(durs)} launches; algorithmic reads {alg / 1e6:.1f} MB -> {e['achieved_GBs']:.0f} GB/s = {e['frac_of_8TBs']:.3f} of 8 TB/s; "
          f"PMC FETCH x2 {2 * (cnt['fetch'] or 0) * 1024 / 1e6:.1f} MB, WRITE {(cnt['write'] or 0) * 1024 / 1e6:.1f} MB"
          + (f" -> {e['hbm_GBs_moved']:.0f} GB/s moved" if 'hbm_GBs_moved' in e else ""))
json.dump(res, open(os.path.join(out, 'gather_traffic.json'), 'w'), indent=1)
