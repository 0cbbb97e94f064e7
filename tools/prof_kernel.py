"""Run one engine kernel a few times (for rocprofv3 --pmc passes).  Usage: python tools/prof_kernel.py NAME [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from news_recommendation_amd import _capi, synth
from news_recommendation_amd._capi import *
name = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device('cuda:0'); lib = _capi.load(); st = lambda: torch.cuda.current_stream().cuda_stream
V = 70976; T = B * 53
g = torch.Generator().manual_seed(0)
table = torch.randn(V, NR_D, generator=g).mul_(0.4).to(dev)
news = synth.news_titles(np.random.default_rng(0), 65238, 20, V)
cand, hist = synth.train_batch(np.random.default_rng(1), news, B)
c, h = synth.batch_token_ids(news, cand, hist)
ids = torch.from_numpy(np.concatenate([c.reshape(-1, 20), h.reshape(-1, 20)])).to(dev)
W = [torch.randn(300, 300, generator=g).mul_(0.06).to(dev) for _ in range(3)]; bb = [torch.randn(300, generator=g).mul_(0.05).to(dev) for _ in range(3)]
Wa = torch.randn(200, 300, generator=g).mul_(0.06).to(dev); ba = torch.zeros(200, device=dev); qv = torch.randn(200, generator=g).mul_(0.1).to(dev)
Wp = torch.empty(3 * NR_NP, NR_KP, dtype=torch.int16, device=dev); bp = torch.empty(3 * NR_NP, device=dev)
Wap = torch.empty(NR_QP, NR_KP, dtype=torch.int16, device=dev); bap = torch.empty(NR_QP, device=dev); qvp = torch.empty(NR_QP, device=dev)
ctx = torch.empty(T * 20, NR_KP, dtype=torch.int16, device=dev)
qs = torch.empty_like(ctx); ks = torch.empty_like(ctx); vts = torch.empty(T, 15, 20, 20, dtype=torch.int16, device=dev)
nv = torch.empty(T, NR_D, device=dev); aw = torch.empty(T, 20, device=dev)
ck = lambda rc: _capi.check(lib, rc)
ck(lib.nr_pack_qkv(W[0].data_ptr(), bb[0].data_ptr(), W[1].data_ptr(), bb[1].data_ptr(), W[2].data_ptr(), bb[2].data_ptr(), Wp.data_ptr(), bp.data_ptr(), st()))
ck(lib.nr_pack_additive(Wa.data_ptr(), ba.data_ptr(), qv.data_ptr(), 200, Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), st()))
xsv = torch.empty_like(ctx)
# training form as the product launches it: Q / K / V^T saves AND the masked token matrix (x_save)
mh = lambda p, save: ck(lib.nr_mhsa_fwd_ex(ids.data_ptr(), table.data_ptr(), V, None, Wp.data_ptr(), bp.data_ptr(), ctx.data_ptr(),
                                           qs.data_ptr() if save else None, ks.data_ptr() if save else None, vts.data_ptr() if save else None,
                                           xsv.data_ptr() if save else None, T, 20, p, 1, st()))
mh(0.0, True)
ck(lib.nr_additive_fwd(ctx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), nv.data_ptr(), aw.data_ptr(), T, 20, st()))
gout = torch.randn(T, NR_D, generator=g).to(dev)
dpre = torch.empty(T * 20, NR_QP, dtype=torch.int16, device=dev); dqp = torch.empty(lib.nr_additive_bwd_grid(T, 20), NR_QP, device=dev)
WaT20 = torch.empty(NR_KP, 224, dtype=torch.int16, device=dev); dctx20 = torch.empty(T * 20, NR_KP, dtype=torch.int16, device=dev)
ck(lib.nr_pack_additive_t(Wa.data_ptr(), 200, WaT20.data_ptr(), st()))
dctx = torch.randn(T * 20, NR_D, generator=g).mul_(0.05).to(dev).to(torch.bfloat16)
dqkv = torch.zeros(T * 20, NR_LDG, dtype=torch.int16, device=dev)
fns = {
  'mhsa_infer': lambda: mh(0.0, False), 'mhsa_train': lambda: mh(0.2, True),
  'additive_fwd': lambda: ck(lib.nr_additive_fwd(ctx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), nv.data_ptr(), aw.data_ptr(), T, 20, st())),
  # as the product launches it: with the fused dctx = dpre @ Wa product (k_pool2.h by default)
  'additive_bwd': lambda: ck(lib.nr_additive_bwd_ex(ctx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), aw.data_ptr(), gout.data_ptr(), dpre.data_ptr(),
                                                    dqp.data_ptr(), WaT20.data_ptr(), dctx20.data_ptr(), T, 20, st())),
  'attn_bwd': lambda: ck(lib.nr_attn_bwd(qs.data_ptr(), ks.data_ptr(), vts.data_ptr(), dctx.data_ptr(), NR_D, aw.data_ptr(), gout.data_ptr(), dqkv.data_ptr(), T, 20, 0.2, 1, st())),
}
if name.startswith(('proj', 'attn_fwd', 'attn_pool', 'attn_bwd_hm', 'dx_gemm', 'tn_gemm')):      # split training forward (csrc/k_proj.h)
    Wp32 = torch.empty(3 * NR_NP * NR_K16 * 16, dtype=torch.int16, device=dev); bp32 = torch.empty(3 * NR_NP, device=dev)
    ck(lib.nr_pack_qkv32(W[0].data_ptr(), bb[0].data_ptr(), W[1].data_ptr(), bb[1].data_ptr(), W[2].data_ptr(), bb[2].data_ptr(), Wp32.data_ptr(), bp32.data_ptr(), st()))
    qkv = torch.empty(T * NR_QKV_HM_SEQ, dtype=torch.int16, device=dev)
    pj = lambda p, xs: ck(lib.nr_qkv_proj_fwd(ids.data_ptr(), table.data_ptr(), V, Wp32.data_ptr(), bp32.data_ptr(), qkv.data_ptr(), xsv.data_ptr() if xs else None, T, 20, p, 1, st()))
    pj(0.2, True)
    dctxk = torch.randn(T * 20, NR_KP, generator=g).mul_(0.05).to(torch.bfloat16).to(dev)
    fns['proj_train'] = lambda: pj(0.2, True)
    fns['proj_noxs'] = lambda: pj(0.2, False)
    fns['attn_fwd'] = lambda: ck(lib.nr_attn_fwd(qkv.data_ptr(), ctx.data_ptr(), None, T, 20, 0.2, 1, st()))
    fns['attn_pool_fwd'] = lambda: ck(lib.nr_attn_pool_fwd(qkv.data_ptr(), ctx.data_ptr(), None, Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), nv.data_ptr(),
                                                           NR_D, aw.data_ptr(), T, 20, 20, 0.2, 1, st()))
    WdX = torch.empty(60 * 10 * 64 * 8, dtype=torch.int16, device=dev)
    ck(lib.nr_pack_qkv_dx(W[0].data_ptr(), W[1].data_ptr(), W[2].data_ptr(), WdX.data_ptr(), st()))
    dX = torch.empty(T * 20, NR_KP, dtype=torch.int16, device=dev)
    z16 = torch.zeros(64, dtype=torch.int16, device=dev)
    P = lib.nr_tn_gemm_parts(NR_LDG, T * 20)
    parts = torch.empty(P, NR_LDG, NR_KP, device=dev)
    fns['dx_gemm'] = lambda: ck(lib.nr_dx_gemm(dqkv.data_ptr(), WdX.data_ptr(), dX.data_ptr(), T * 20, st()))
    fns['tn_gemm'] = lambda: ck(lib.nr_tn_gemm(dqkv.data_ptr(), NR_LDG, NR_LDG, xsv.data_ptr(), z16.data_ptr(), parts.data_ptr(), T * 20, P, st()))
    fns['attn_bwd_hm'] = lambda: ck(lib.nr_attn_bwd_hm(qkv.data_ptr(), dctxk.data_ptr(), NR_KP, aw.data_ptr(), gout.data_ptr(), dqkv.data_ptr(), None, T, 20, 0.2, 1, st()))
if name in ('dx_gemm', 'tn_gemm'):        # a realistic dqkv operand: the attention backward's output
    fns['attn_bwd_hm']()
if name.endswith('50'):          # abstract-shaped pooling: 27 k sequences of 50 ctx rows
    S = 50; Tn = B * 53
    ctx50 = torch.randn(Tn * S, NR_KP, generator=g).mul_(0.3).to(torch.bfloat16).view(torch.int16).to(dev)
    nv50 = torch.empty(Tn, NR_D, device=dev); aw50 = torch.empty(Tn, S, device=dev)
    ck(lib.nr_additive_fwd(ctx50.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), nv50.data_ptr(), aw50.data_ptr(), Tn, S, st()))
    gout50 = torch.randn(Tn, NR_D, generator=g).to(dev)
    dpre50 = torch.empty(Tn * S, NR_QP, dtype=torch.int16, device=dev); dqp50 = torch.empty(lib.nr_additive_bwd_grid(Tn, S), NR_QP, device=dev)
    WaT = torch.empty(NR_KP, 224, dtype=torch.int16, device=dev); dctx50 = torch.empty(Tn * S, NR_KP, dtype=torch.int16, device=dev)
    ck(lib.nr_pack_additive_t(Wa.data_ptr(), 200, WaT.data_ptr(), st()))
    fns['additive_fwd50'] = lambda: ck(lib.nr_additive_fwd(ctx50.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), nv50.data_ptr(), aw50.data_ptr(), Tn, S, st()))
    fns['additive_bwd50'] = lambda: ck(lib.nr_additive_bwd_ex(ctx50.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), aw50.data_ptr(), gout50.data_ptr(),
                                                              dpre50.data_ptr(), dqp50.data_ptr(), WaT.data_ptr(), dctx50.data_ptr(), Tn, S, st()))
if name.startswith('conv'):
    S = 50 if 'abs' in name else 20
    Tn = B * 53
    idc = torch.randint(1, V, (Tn, S), generator=g).to(dev)
    Wcv = torch.randn(300, 1, 3, 300, generator=g).mul_(0.03).to(dev); bcv = torch.zeros(300, device=dev)
    Wc = torch.empty(3, NR_KP, NR_KP, dtype=torch.int16, device=dev); Wd = torch.empty_like(Wc); bc = torch.empty(NR_KP, device=dev)
    ck(lib.nr_pack_conv(Wcv.data_ptr(), bcv.data_ptr(), 300, 300, Wc.data_ptr(), Wd.data_ptr(), bc.data_ptr(), st()))
    act = torch.empty(Tn * S, NR_KP, dtype=torch.int16, device=dev)
    xs = torch.empty(Tn * (S + 1) + 1, NR_KP, dtype=torch.int16, device=dev)
    fns[name] = lambda: ck(lib.nr_conv3_fwd(idc.data_ptr(), table.data_ptr(), V, Wc.data_ptr(), bc.data_ptr(), act.data_ptr(), xs.data_ptr(), Tn, S, 0.2, 1, 0, st()))
if name.startswith('pool_flat'):     # flat pooling backward (csrc/k_pool3.h): pool_flat / pool_flat_act (titles), pool_flat50 / pool_flat50_act (abstracts)
    S = 50 if '50' in name else 20
    Tn = B * 53
    cx = torch.randn(Tn * S, NR_KP, generator=g).mul_(0.3)
    if 'act' in name: cx = torch.relu(cx)
    cx = cx.to(torch.bfloat16).view(torch.int16).to(dev)
    y_ = torch.empty(Tn, NR_D, device=dev); aw_ = torch.empty(Tn, S, device=dev)
    ck(lib.nr_additive_fwd(cx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), y_.data_ptr(), aw_.data_ptr(), Tn, S, st()))
    go_ = torch.randn(Tn, NR_D, generator=g).to(dev)
    dp_ = torch.empty(Tn * S, NR_QP, dtype=torch.int16, device=dev); dq_ = torch.empty(lib.nr_additive_bwd_flat_grid(Tn * S), NR_QP, device=dev)
    tot_ = torch.empty(Tn, device=dev)
    dc_ = torch.empty(Tn * S, NR_KP, dtype=torch.int16, device=dev)
    dy_ = torch.zeros(Tn * (S + 1) + 1, NR_KP, dtype=torch.int16, device=dev)
    act_ = 'act' in name
    gs_ = NR_D
    if 'gs' in name:                    # the sequence gradients as the last third of [Tn, 900] rows (LSTUR's news-vector gradient): nr_additive_bwd_flat_gs
        wide_ = torch.zeros(Tn, 3 * NR_D, device=dev); wide_[:, 2 * NR_D:] = go_; go_ = wide_[:, 2 * NR_D:]; gs_ = 3 * NR_D
    fns[name] = lambda: ck(lib.nr_additive_bwd_flat_gs(cx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), aw_.data_ptr(), go_.data_ptr(), gs_, y_.data_ptr(), NR_D,
                                                       tot_.data_ptr(), dp_.data_ptr(), dq_.data_ptr(), None if act_ else dc_.data_ptr(), dy_.data_ptr() if act_ else None,
                                                       0.2 if act_ else 0.0, Tn, S, 200, st()))
if name.startswith('cgemm'):          # the convolution as a persistent ring GEMM (csrc/k_convgemm.h): cgemm_dgrad50 / cgemm_dgrad20; NR_CONVGEMM_DEBUG switches phases off, NR_CONV_GEMM_PERSIST=0: the one-tile-per-workgroup kernel
    S = 50 if '50' in name else 20
    Tn = B * 55
    Wcv = torch.randn(300, 1, 3, 300, generator=g).mul_(0.03).to(dev)
    Wd2 = torch.empty(NR_KP, 3 * NR_KP, dtype=torch.int16, device=dev)
    ck(lib.nr_pack_conv_dgrad(Wcv.data_ptr(), 300, 300, Wd2.data_ptr(), st()))
    if int(os.environ.get('NR_CONVGEMM_DEBUG', '0')) & 16:      # experiment: the filter bank chunk-major [30 chunks = (column block, tap)][KP rows][32]
        Wd2 = Wd2.view(NR_KP, 3, 10, 32).permute(2, 1, 0, 3).contiguous()
    dyp = torch.randn(Tn * (S + 1) + 1, NR_KP, generator=g).mul_(0.1).to(torch.bfloat16).view(torch.int16).to(dev)
    dxo = torch.empty(Tn * S, NR_KP, dtype=torch.int16, device=dev)
    fns[name] = lambda: ck(lib.nr_conv3_dgrad_gemm(dyp.data_ptr(), Wd2.data_ptr(), dxo.data_ptr(), Tn, S, st()))
if name.startswith('pool_fwd_flat'):     # whole-sequence pooling forward (csrc/k_pool4.h): pool_fwd_flat (titles) / pool_fwd_flat50 (abstracts); NR_POOL_DEBUG switches phases off
    S = 50 if '50' in name else 20
    Tn = B * (55 if S == 50 else 55)
    cx = torch.relu(torch.randn(Tn * S, NR_KP, generator=g).mul_(0.3)).to(torch.bfloat16).view(torch.int16).to(dev)
    y_ = torch.empty(Tn, NR_D, device=dev); aw_ = torch.empty(Tn, S, device=dev)
    fns[name] = lambda: ck(lib.nr_additive_fwd_flat(cx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), y_.data_ptr(), NR_D, None, 0, aw_.data_ptr(), Tn, S, S, 200, st()))
fn = fns[name]
for _ in range(2): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): fn()
e1.record(); torch.cuda.synchronize()
print(name, 'avg_us', e0.elapsed_time(e1) / 5 * 1e3)
