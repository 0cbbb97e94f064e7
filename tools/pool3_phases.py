"""Phase decomposition of the flat pooling backward (csrc/k_pool3.h): the DBG instantiation with one phase switched off at a time, all in one
process (NR_POOL_DEBUG is re-read by every call).  Bits: 1 no ctx loads, 2 no dw phase, 4 no projection MFMAs, 8 no tanh / dpre / dq arithmetic,
16 no dctx product, 32 no global stores, 128 no dq accumulation; 64 = nothing off (the DBG build's own baseline).
Usage: python tools/pool3_phases.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from news_recommendation_amd import _capi
from news_recommendation_amd._capi import NR_D, NR_KP, NR_QP

TIMELINE = '--timeline' in sys.argv
PROD_ONLY = '--prod-only' in sys.argv
args = [a for a in sys.argv[1:] if not a.startswith('--')]
B = int(args[0]) if args else 512
dev = torch.device('cuda:0'); lib = _capi.load(); st = lambda: torch.cuda.current_stream().cuda_stream
ck = lambda rc: _capi.check(lib, rc)
g = torch.Generator().manual_seed(0)
Wa = torch.randn(200, 300, generator=g).mul_(0.06).to(dev); ba = torch.zeros(200, device=dev); qv = torch.randn(200, generator=g).mul_(0.1).to(dev)
Wap = torch.empty(NR_QP, NR_KP, dtype=torch.int16, device=dev); bap = torch.empty(NR_QP, device=dev); qvp = torch.empty(NR_QP, device=dev)
ck(lib.nr_pack_additive(Wa.data_ptr(), ba.data_ptr(), qv.data_ptr(), 200, Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), st()))


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for S, act, Tn in ((20, False, B * 53), (20, True, B * 55), (50, True, B * 55), (50, False, 512), (50, False, B * 55 // 2)):
    cx = torch.randn(Tn * S, NR_KP, generator=g).mul_(0.3)
    if act: cx = torch.relu(cx)
    cx = cx.to(torch.bfloat16).view(torch.int16).to(dev)
    y_ = torch.empty(Tn, NR_D, device=dev); aw_ = torch.empty(Tn, S, device=dev)
    ck(lib.nr_additive_fwd(cx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), y_.data_ptr(), aw_.data_ptr(), Tn, S, st()))
    go_ = torch.randn(Tn, NR_D, generator=g).to(dev)
    dp_ = torch.empty(Tn * S, NR_QP, dtype=torch.int16, device=dev); dq_ = torch.empty(lib.nr_additive_bwd_flat_grid(Tn * S), NR_QP, device=dev)
    tot_ = torch.empty(Tn, device=dev)
    dc_ = torch.empty(Tn * S, NR_KP, dtype=torch.int16, device=dev)
    dy_ = torch.zeros(Tn * (S + 1) + 1, NR_KP, dtype=torch.int16, device=dev)
    fn = lambda: ck(lib.nr_additive_bwd_flat(cx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), aw_.data_ptr(), go_.data_ptr(), y_.data_ptr(), NR_D,
                                             tot_.data_ptr(), dp_.data_ptr(), dq_.data_ptr(), None if act else dc_.data_ptr(), dy_.data_ptr() if act else None,
                                             0.2 if act else 0.0, Tn, S, 200, st()))
    if TIMELINE:
        # cycle-counter stamps of the first 8 iterations of waves 0-1 of workgroups 0-3 (debug build, nothing switched off): 0 top of the
        # iteration, 1 rows are fragments, 2 ds known, 3 six n-tiles done, 4 projection done, 5 next rows requested, 6 dctx done, 7 end
        import numpy as np
        st_buf = torch.zeros(4 * 2 * 8 * 8, dtype=torch.int64, device=dev)
        lib.nr_debug_pool3_stamps(st_buf.data_ptr())
        os.environ['NR_POOL_DEBUG'] = '64'
        fn(); torch.cuda.synchronize()
        lib.nr_debug_pool3_stamps(None)
        a = st_buf.cpu().numpy().reshape(4, 2, 8, 8)
        print(f"timeline S={S} act={int(act)} (ticks; columns: to-fragments, dw, n-tiles 0-5, n-tiles 6-12, prefetch issue, dctx, tail | iteration total)")
        for wg in range(2):
            for wv in range(2):
                for it in range(6):
                    r = a[wg, wv, it]
                    if r[0] == 0: continue
                    d = [int(r[k + 1] - r[k]) for k in range(7)]
                    nxt = int(a[wg, wv, it + 1][0] - r[0]) if it + 1 < 8 and a[wg, wv, it + 1][0] else -1
                    print(f"  wg{wg} w{wv} it{it}: start {int(r[0] - a[0, 0, 0][0]):8d} | " + ' '.join(f'{x:6d}' for x in d) + f" | {nxt}")
    os.environ.pop('NR_POOL_DEBUG', None)
    line = [f"S={S} act={int(act)} n_seq={Tn} | production {timed(fn):.1f}"]
    for d in (() if PROD_ONLY else (64, 65, 66, 68, 72, 80, 96, 192, 48, 255)):
        os.environ['NR_POOL_DEBUG'] = str(d)
        line.append(f"{d}: {timed(fn):.1f}")
    print(' | '.join(line), flush=True)
