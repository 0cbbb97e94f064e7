"""Phase switches and per-wave timelines of nr_attn_bwd_hm, both forms in one process: the DMA form (csrc/k_attn_bwd2.h, NR_ATTNB2=1, default; its five-wave variant of the round's A/B is no longer instantiated) and
the round-4 TILE form (csrc/k_bwd.h, NR_ATTNB2=0).  Cycle-counter stamps of the first titles of workgroups 0-1, every wave and round.
Usage: python tools/attnb_timeline.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from news_recommendation_amd import _capi
from news_recommendation_amd._capi import NR_D, NR_KP, NR_LDG, NR_QKV_HM_SEQ

args = [a for a in sys.argv[1:] if not a.startswith('--')]
B = int(args[0]) if args else 512
dev = torch.device('cuda:0'); lib = _capi.load(); st = lambda: torch.cuda.current_stream().cuda_stream
ck = lambda rc: _capi.check(lib, rc)
g = torch.Generator().manual_seed(0)
T = B * 53
ntok = T * 20
qkv = torch.randn(T * NR_QKV_HM_SEQ, generator=g).mul_(0.5).to(torch.bfloat16).view(torch.int16).to(dev)
dctx = torch.randn(ntok, NR_KP, generator=g).mul_(0.05).to(torch.bfloat16).to(dev)
aw = torch.full((T, 20), 0.05, device=dev); go = torch.randn(T, NR_D, generator=g).to(dev)
dqkv = torch.zeros(ntok, NR_LDG, dtype=torch.int16, device=dev)
fn = lambda: ck(lib.nr_attn_bwd_hm(qkv.data_ptr(), dctx.data_ptr(), NR_KP, aw.data_ptr(), go.data_ptr(), dqkv.data_ptr(), None, T, 20, 0.2, 1, st()))


def timed(n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3




def phases(tag):
    os.environ.pop('NR_ATTNB_DEBUG', None)
    print(f"[{tag}] production {timed():.1f} us")
    for d, what in ((8, 'nothing off'), (9, 'no global loads'), (12, 'no dqkv stores'), (13, 'no loads, no stores'), (24, 'no arithmetic (I/O skeleton)'),
                    (25, 'no arithmetic, no loads'), (28, 'no arithmetic, no stores'), (29, 'barriers and LDS staging only')):
        os.environ['NR_ATTNB_DEBUG'] = str(d)
        print(f"[{tag}] debug build {d:2d} {what:32s} {timed():.1f} us", flush=True)


def timeline(NW, NR, names, title_cols):
    os.environ['NR_ATTNB_DEBUG'] = '8'
    NS, NK = 4, 12
    buf = torch.zeros(2 * NW * NS * NR * NK, dtype=torch.int64, device=dev)
    lib.nr_debug_attnb_stamps(buf.data_ptr())
    fn(); torch.cuda.synchronize()
    print(f"debug build with stamps {timed(3):.1f} us")
    buf.zero_()
    fn(); torch.cuda.synchronize()
    lib.nr_debug_attnb_stamps(None)
    a = buf.cpu().numpy().reshape(2, NW, NS, NR, NK)
    t0 = a[a > 0].min()
    print("ticks (s_memtime = shader clock); columns: start | " + ' '.join(f'{n:>9s}' for n in names))
    for wg in range(2):
        for it in range(NS):
            for rnd in range(NR):
                for wv in range(NW):
                    r = a[wg, wv, it, rnd]
                    ks = [k for k in range(NK) if r[k]]
                    if not ks:
                        continue
                    d = [(int(r[k + 1] - r[k]) if (r[k + 1] and r[k]) else -1) for k in range(len(names))]
                    print(f"wg{wg} title{it} rnd{rnd} w{wv}: {int(r[ks[0]] - t0):8d} | " + ' '.join(f'{x:9d}' for x in d))
            print()
    os.environ.pop('NR_ATTNB_DEBUG', None)


# ---- interleaved A/B of the production builds (the first timed launches of a process run at a lower clock: warm up first) ---------------------
for _ in range(40): fn()
torch.cuda.synchronize()
variants = {'TILE (r04)': {'NR_ATTNB2': '0'}, 'DMA (8 waves)': {'NR_ATTNB2': '1'}}
res = {k: [] for k in variants}
for rnd_ in range(5):
    for k, env in variants.items():
        os.environ.update(env)
        res[k].append(timed(10))
for k, v in res.items():
    print(f"A/B {k:12s} median {sorted(v)[len(v) // 2]:7.1f} us   min {min(v):7.1f}   all {' '.join(f'{x:.0f}' for x in v)}", flush=True)
if '--ab-only' in sys.argv:
    sys.exit(0)

os.environ['NR_ATTNB2'] = '1'
phases('DMA form')
# stamps of the DMA form: 0 title top, 1 write-out issued, 2 dC pass done, (barrier), 3 pair top, 4 fragments read + copies issued, 5 P / dP / dS,
# 6 transposes, 7 outputs in the tile, 8 in front of the title barrier (round 2), 9 behind it
timeline(8, 2, ['writeout', 'dC pass', 'barrier B', 'frags+dma', 'softmax', 'transp', 'outputs', '(next)', 'barrier C'], None)
os.environ['NR_ATTNB2'] = '0'
phases('TILE form')
timeline(4, 4, ['store_lds', 'prefetch', 'operands', 'softmax', 'transp', 'outputs', '(gap)', 'barrier1', 'writeout', 'barrier2'], None)
