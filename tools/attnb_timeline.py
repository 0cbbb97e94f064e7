"""Per-wave timeline of nr_attn_bwd_hm (csrc/k_bwd.h, TILE form): cycle-counter stamps of the first sequences of workgroups 0-1, every wave and
round.  Stamps: 0 top of the pair, 1 operands in wave-private LDS, 2 next pair requested, 3 fragments / transposed operands ready, 4 P, dP, dS done,
5 P / dS transposed, 6 outputs in the title tile, 7 in front of the title barrier, 8 behind it, 9 write-out issued, 10 behind the second barrier.
Usage: python tools/attnb_timeline.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from news_recommendation_amd import _capi
from news_recommendation_amd._capi import NR_D, NR_KP, NR_LDG, NR_QKV_HM_SEQ

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
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


os.environ.pop('NR_ATTNB_DEBUG', None)
print(f"production {timed():.1f} us")
for d, what in ((8, 'nothing off'), (9, 'no global loads'), (12, 'no dqkv stores'), (13, 'no loads, no stores'), (24, 'no arithmetic (I/O skeleton)'),
                (25, 'no arithmetic, no loads'), (28, 'no arithmetic, no stores'), (29, 'barriers and LDS staging only')):
    os.environ['NR_ATTNB_DEBUG'] = str(d)
    print(f"debug build {d:2d} {what:32s} {timed():.1f} us", flush=True)
os.environ['NR_ATTNB_DEBUG'] = '8'
NW, NS, NR, NK = 4, 4, 4, 12
buf = torch.zeros(2 * NW * NS * NR * NK, dtype=torch.int64, device=dev)
lib.nr_debug_attnb_stamps(buf.data_ptr())
fn(); torch.cuda.synchronize()
print(f"debug build with stamps {timed(3):.1f} us")
buf.zero_()
fn(); torch.cuda.synchronize()
lib.nr_debug_attnb_stamps(None)
a = buf.cpu().numpy().reshape(2, NW, NS, NR, NK)
t0 = a[a > 0].min()
names = ['store_lds', 'prefetch', 'operands', 'softmax', 'transp', 'outputs', '(gap)', 'barrier1', 'writeout', 'barrier2']
print("ticks (s_memtime = shader clock); columns: start | " + ' '.join(f'{n:>9s}' for n in names) + " | pair total")
for wg in range(2):
    for it in range(NS):
        for rnd in range(NR):
            for wv in range(NW):
                r = a[wg, wv, it, rnd]
                if r[0] == 0:
                    continue
                d = []
                for k in range(10):
                    d.append(int(r[k + 1] - r[k]) if r[k + 1] and r[k] else -1)
                last = max(int(x) for x in r if x)
                print(f"wg{wg} seq{it} rnd{rnd} w{wv}: {int(r[0] - t0):8d} | " + ' '.join(f'{x:9d}' for x in d) + f" | {last - int(r[0])}")
        print()
