"""Persistent (XCD-local) GRU sweeps against the step-per-launch form: parity through tests/kernel_checks_gru.check_gru (which compares the sweep
entry points bit for bit with the step launches and both with the float64 oracle), the error words, and an interleaved timing A/B at the LSTUR
launch size (B = 512, N = 50, Hd = 900).  Usage: python tools/gru_persist_ab.py [--no-check]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
from news_recommendation_amd import _capi

lib = _capi.load()
dev = torch.device('cuda:0')
st = lambda: torch.cuda.current_stream().cuda_stream
ck = lambda rc: _capi.check(lib, rc)


def status():
    a, b = ctypes.c_int32(0), ctypes.c_int32(0)
    ck(lib.nr_gru_persist_status(ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


if '--no-check' not in sys.argv:
    from tests.backends import GpuBackend
    from tests import kernel_checks_gru as kcg
    be = GpuBackend()
    for kw in (dict(B=512, N=12, Hd=900, I=900, seed=6), dict(B=133, N=50, Hd=900, I=900), dict(B=17, N=6, Hd=900, I=900),
               dict(B=512, N=9, Hd=450, I=900, seed=7, lens=[1 + (7 * i) % 9 for i in range(512)]), dict(B=70, N=20, Hd=450, I=900, seed=1)):
        err = kcg.check_gru(be, **kw)
        print('check_gru', {k: v for k, v in kw.items() if k != 'lens'}, f'fwd err {err:.3g}', 'status', status(), flush=True)

B, N, Hd = 512, 50, 900
Hg, Hp = (Hd + 15) // 16 * 16, (Hd + 1 + 31) // 32 * 32
Kp = (3 * Hg + 31) // 32 * 32
g = torch.Generator().manual_seed(0)
gi = torch.randn(B * N, 3 * Hg, generator=g).mul_(0.5).to(dev)
Whh = torch.randn(3 * Hg, Hp, generator=g).mul_(0.03).to(torch.bfloat16).view(torch.int16).to(dev)
WhhT = torch.randn(Hp, Kp, generator=g).mul_(0.03).to(torch.bfloat16).view(torch.int16).to(dev)
b_ih = torch.zeros(3 * Hd, device=dev); b_hh = torch.zeros(3 * Hd, device=dev)
ln = torch.randint(1, N + 1, (B,), generator=g).to(torch.int32); ln[0] = N
ln = ln.to(dev)
ht2 = torch.zeros(2, B, Hp, dtype=torch.int16, device=dev); hf2 = torch.zeros(2, B, Hp, device=dev)
H_all = torch.zeros(N + 1, B, Hp, dtype=torch.int16, device=dev); gates = torch.zeros(N, B, 4, Hg, dtype=torch.int16, device=dev)
glast = torch.randn(B, Hd, generator=g).to(dev)
dgi = torch.zeros(B * N, Kp, dtype=torch.int16, device=dev); dgh = torch.zeros(N, B, Kp, dtype=torch.int16, device=dev)
dght2 = torch.zeros(2, B, Kp, dtype=torch.int16, device=dev); carry2 = torch.zeros(2, B, Hp, device=dev)
fwd = lambda: ck(lib.nr_gru_fwd_seq(gi.data_ptr(), Whh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), ln.data_ptr(), ht2.data_ptr(), H_all.data_ptr(),
                                    hf2.data_ptr(), gates.data_ptr(), B, N, Hd, N, st()))
bwd = lambda: ck(lib.nr_gru_bwd_seq(glast.data_ptr(), WhhT.data_ptr(), gates.data_ptr(), H_all.data_ptr(), ln.data_ptr(), dgi.data_ptr(), dgh.data_ptr(),
                                    dght2.data_ptr(), carry2.data_ptr(), B, N, Hd, N, st()))


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for _ in range(10): fwd(); bwd()
torch.cuda.synchronize()
res = {}
for rnd in range(4):
    for name, env in (('steps', '0'), ('persistent', '3')):
        os.environ['NR_GRU_PERSIST'] = env
        res.setdefault(name + ' fwd', []).append(timed(fwd))
        res.setdefault(name + ' bwd', []).append(timed(bwd))
for k, v in res.items():
    print(f'{k:16s} median {sorted(v)[len(v) // 2]:8.1f} us   min {min(v):8.1f}   ({" ".join(f"{x:.0f}" for x in v)})')
print('status', status())

if '--timeline' in sys.argv:
    os.environ['NR_GRU_PERSIST'] = '3'
    buf = torch.zeros(256 * N * 8 * 8, dtype=torch.int64, device=dev)
    lib.nr_debug_gru_stamps(buf.data_ptr())
    fwd(); torch.cuda.synchronize()
    buf.zero_(); fwd(); torch.cuda.synchronize()
    lib.nr_debug_gru_stamps(None)
    a = buf.cpu().numpy().reshape(256, N, 8, 8).astype(np.int64)
    a = (a - a[a > 0].min()) * 10                      # ns
    names = ['issue', 'dma wait', 'sync1', 'mfma', 'sync2', 'gates', 'drain', 'barrier']
    print('ns (s_memrealtime, 10 ns resolution); workgroup 0; columns: step start | ' + ' '.join(f'{n:>8s}' for n in names))
    for t in (10, 30):
        for w in range(8):
            r = a[0][t]
            d = [int(r[w][k + 1] - r[w][k]) for k in range(7)] + [int(a[0][t + 1][w][0] - r[w][7])]
            print(f't{t} w{w}: {int(r[w][0] - a[0][10][0][0]):7d} | ' + ' '.join(f'{x:8d}' for x in d))
        print()
    # per step, over the 32 workgroups of XCD 0 (blockIdx % 8 == 0): when does each arrive at the inter-workgroup wait (latest wave), and which phase
    # made the latest one late
    wg = np.arange(0, 256, 8)
    arr = a[wg][:, :, :, 7].max(axis=2)                 # [32][T] arrival = latest wave's drain end
    top = a[wg][:, :, :, 0].min(axis=2)
    print('XCD 0, per step: spread of arrivals at the wait (latest - earliest, ns), (latest - median), step length')
    for t in range(5, N - 1, 4):
        late = int(arr[:, t].argmax())
        ph = a[wg[late], t]                             # [8 waves][8]
        med = a[wg][:, t]                               # [32][8][8]
        dur = lambda x: np.diff(x, axis=-1).max(axis=-2)        # per phase, slowest wave
        d_late, d_med = dur(ph), np.median(dur(med), axis=0)
        print(f't{t:2d}: spread {int(arr[:, t].max() - arr[:, t].min()):6d}  late-median {int(arr[:, t].max() - np.median(arr[:, t])):6d}  step {int(top[:, t + 1].min() - top[:, t].min()):6d}'
              f'  latest = slot-wg {late:2d}; its phases {" ".join(f"{int(x):5d}" for x in d_late)} | median {" ".join(f"{int(x):5d}" for x in d_med)}')

if '--timeline-bwd' in sys.argv:
    os.environ['NR_GRU_PERSIST'] = '3'
    NC = N + 1
    buf = torch.zeros(256 * NC * 8 * 8, dtype=torch.int64, device=dev)
    fwd(); lib.nr_debug_gru_stamps_bwd(buf.data_ptr())
    bwd(); torch.cuda.synchronize()
    buf.zero_(); bwd(); torch.cuda.synchronize()
    lib.nr_debug_gru_stamps_bwd(None)
    a = buf.cpu().numpy().reshape(256, NC, 8, 8).astype(np.int64)
    a = (a - a[a > 0].min()) * 10
    names = ['loads+mfma', 'exch', 'sync', 'epilogue', 'drain', 'barrier']
    print('backward sweep, ns; workgroup 0; columns: call start | ' + ' '.join(f'{n:>10s}' for n in names))
    for i in (10, 30):
        for w in range(8):
            r = a[0][i][w]
            d = [int(r[k + 1] - r[k]) for k in range(5)] + [int(a[0][i + 1][w][0] - r[5])]
            print(f'i{i} w{w}: {int(r[0] - a[0][10][0][0]):7d} | ' + ' '.join(f'{x:10d}' for x in d))
        print()
    wg = np.arange(0, 256, 8)
    top = a[wg][:, :, :, 0].min(axis=2)
    arr = a[wg][:, :, :, 5].max(axis=2)
    for i in range(5, N - 1, 8):
        print(f'call {i:2d}: length {int(top[:, i + 1].min() - top[:, i].min()):6d} ns, arrival spread {int(arr[:, i].max() - arr[:, i].min()):6d}, latest - median {int(arr[:, i].max() - np.median(arr[:, i])):6d}')
