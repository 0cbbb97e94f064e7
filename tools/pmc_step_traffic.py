"""HBM traffic of a WHOLE training step, kernel by kernel, from two rocprofv3 passes over `bench.py --no-graph` (FETCH_SIZE and WRITE_SIZE in
separate runs, MI355X_MICROARCH.md): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch, summed per kernel name over all dispatches and
divided by the number of steps the command ran (warm-up + the two profiled steps + the timed ones, all identical steps).
Usage: python tools/pmc_step_traffic.py OUTDIR TAG N_STEPS  (expects OUTDIR/step_<TAG>_{fetch,write}/ with counter_collection CSVs).
Writes OUTDIR/step_traffic_<TAG>.txt and merges {TAG: {...}} into OUTDIR/step_traffic.json (what bench.py reports as traffic_total_per_step)."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
out, tag, n_steps = sys.argv[1], sys.argv[2], int(sys.argv[3])


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:90]


tot = {}
for c, col in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    for f in glob.glob(f'{out}/step_{tag}_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != col:
                continue
            e = tot.setdefault(short(r['Kernel_Name']), {'fetch': 0.0, 'write': 0.0, 'n': 0})
            e[c] += float(r['Counter_Value'])
            if c == 'fetch':
                e['n'] += 1
rows = []
for k, e in tot.items():
    b = (2 * e['fetch'] + e['write']) * 1024 / n_steps
    rows.append((b, k, e['n'] / n_steps, 2 * e['fetch'] * 1024 / n_steps, e['write'] * 1024 / n_steps))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
lines = [f"HBM traffic per training step ({tag}), rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE (KiB units), summed over all dispatches / {n_steps} steps",
         f"TOTAL {total / 1e9:.3f} GB per step", "", f"{'GB/step':>9} {'fetch x2':>9} {'write':>8} {'launches':>8}  kernel"]
for b, k, n, f2, w in rows[:45]:
    lines.append(f"{b / 1e9:9.3f} {f2 / 1e9:9.3f} {w / 1e9:8.3f} {n:8.1f}  {k}")
open(f'{out}/step_traffic_{tag}.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:30]))
jf = f'{out}/step_traffic.json'
d = json.load(open(jf)) if os.path.exists(jf) else {}
d[tag] = {"bytes_per_step": int(total), "source_hash": bench.kernel_source_hash(), "steps": n_steps,
          "top": [[k, int(b)] for b, k, *_ in rows[:12]]}
json.dump(d, open(jf, 'w'), indent=1)
