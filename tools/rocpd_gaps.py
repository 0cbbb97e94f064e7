"""Idle-gap analysis of a rocprofv3 rocpd database: kernels in start order, GPU busy fraction over the steady-state window and the
largest gaps with the kernels around them (is the step GPU-bound or is the host starving the queue?).  Usage: python tools/rocpd_gaps.py DB"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
print("columns:", cols)
sc = 'start' if 'start' in cols else [x for x in cols if 'start' in x][0]
ec = 'end' if 'end' in cols else [x for x in cols if 'end' in x][0]
extra = [x for x in ('stream_id', 'queue_id', 'tid') if x in cols]
rows = c.execute(f"select name, {sc}, {ec}" + ''.join(', ' + x for x in extra) + f" from kernels order by {sc}").fetchall()
n = len(rows)
rows = rows[n // 3:]                      # steady state: drop the first third (warm-up, allocation)
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy, cur_end, gaps = 0, rows[0][1], []
for i, r in enumerate(rows):
    s, e = r[1], r[2]
    if s > cur_end:
        gaps.append((s - cur_end, rows[i - 1][0][:60] if i else '', r[0][:60], r[3:] ))
        busy += e - s
    else:
        busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
print(f"window {(t1 - t0) / 1e6:.2f} ms, kernels {len(rows)}, busy (union) {busy / 1e6:.2f} ms = {100.0 * busy / (t1 - t0):.1f} %")
tot_gap = sum(g[0] for g in gaps)
print(f"gaps: {len(gaps)}, total {tot_gap / 1e6:.2f} ms; > 20 us: {sum(1 for g in gaps if g[0] > 20000)}")
agg = {}
for g in gaps:
    k = (g[1], g[2])
    a = agg.setdefault(k, [0, 0])
    a[0] += g[0]; a[1] += 1
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{a[0] / 1e3:9.1f} us in {a[1]:4d} gaps | after [{k[0]}] before [{k[1]}]")
