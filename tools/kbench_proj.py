"""Micro-benchmark of the split training forward (csrc/k_proj.h) beside the register-resident kernel, on cuda:0 (HIP events on the launch
stream, MIND-small shape, B = 512 impressions = 27,136 titles).  Usage: python tools/kbench_proj.py [B]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from news_recommendation_amd import _capi
from news_recommendation_amd._capi import NR_D, NR_KP, NR_NP, NR_LDG, NR_QKV_HM_SEQ, NR_K16, NR_QP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
V = 70976
dev = torch.device('cuda:0')
lib = _capi.load()
st = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device='cpu').manual_seed(0)
T = B * 53
table = torch.randn(V, NR_D, generator=g).to(dev)
rng = np.random.default_rng(0)
ids_np = np.minimum(rng.zipf(1.2, size=(T, 20)), V - 1).astype(np.int64)
ids_np[:, 11:] = 0
ids = torch.from_numpy(ids_np).to(dev)
W = [torch.randn(300, 300, generator=g).mul_(0.05).to(dev) for _ in range(3)]
bb = [torch.randn(300, generator=g).mul_(0.05).to(dev) for _ in range(3)]
Wp = torch.empty(3 * NR_NP, NR_KP, dtype=torch.int16, device=dev); bp = torch.empty(3 * NR_NP, device=dev)
Wp32 = torch.empty(3 * NR_NP * NR_K16 * 16, dtype=torch.int16, device=dev); bp32 = torch.empty(3 * NR_NP, device=dev)
ck = lambda rc: _capi.check(lib, rc)
wargs = [W[0].data_ptr(), bb[0].data_ptr(), W[1].data_ptr(), bb[1].data_ptr(), W[2].data_ptr(), bb[2].data_ptr()]
ck(lib.nr_pack_qkv(*wargs, Wp.data_ptr(), bp.data_ptr(), st()))
ck(lib.nr_pack_qkv32(*wargs, Wp32.data_ptr(), bp32.data_ptr(), st()))
ntok = T * 20
ctx = torch.empty(ntok, NR_KP, dtype=torch.int16, device=dev)
qkv = torch.empty(T * NR_QKV_HM_SEQ, dtype=torch.int16, device=dev)
xs = torch.empty(ntok, NR_KP, dtype=torch.int16, device=dev)
qs = torch.empty(ntok, NR_KP, dtype=torch.int16, device=dev); ks = torch.empty(ntok, NR_KP, dtype=torch.int16, device=dev)
vts = torch.empty(T, 15, 20, 20, dtype=torch.int16, device=dev)
dctx = torch.randn(ntok, NR_KP, generator=g).mul_(0.05).to(torch.bfloat16).to(dev)
aw = torch.full((T, 20), 0.05, device=dev); go = torch.randn(T, NR_D, generator=g).to(dev)
dqkv = torch.zeros(ntok, NR_LDG, dtype=torch.int16, device=dev)
p = 0.2
Wa = torch.randn(200, 300, generator=g).mul_(0.05).to(dev); ba = torch.zeros(200, device=dev); qv = torch.randn(200, generator=g).mul_(0.1).to(dev)
Wap = torch.empty(NR_QP, NR_KP, dtype=torch.int16, device=dev); bap = torch.empty(NR_QP, device=dev); qvp = torch.empty(NR_QP, device=dev)
ck(lib.nr_pack_additive(Wa.data_ptr(), ba.data_ptr(), qv.data_ptr(), 200, Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), st()))
nv = torch.empty(T, NR_D, device=dev); awo = torch.empty(T, 20, device=dev)
WdX = torch.empty(60 * 10 * 64 * 8, dtype=torch.int16, device=dev)
ck(lib.nr_pack_qkv_dx(W[0].data_ptr(), W[1].data_ptr(), W[2].data_ptr(), WdX.data_ptr(), st()))
Wall = torch.zeros(NR_LDG, NR_KP, device=dev)
for i in range(3):
    Wall[i * NR_KP:i * NR_KP + 300, :300] = W[i]
WpT = Wall.to(torch.bfloat16).t().contiguous()                       # [KP, 960]: the operand form ops.pack_qkv_t hands to hipBLASLt
dqkv_r = torch.zeros(ntok, NR_LDG, device=dev)
for i in range(3):
    dqkv_r[:, i * NR_KP:i * NR_KP + 300] = torch.randn(ntok, 300, device=dev) * 0.1
dqkv_b = dqkv_r.to(torch.bfloat16); del dqkv_r
dX_hand = torch.empty(ntok, NR_KP, dtype=torch.bfloat16, device=dev)
from news_recommendation_amd import ops
from news_recommendation_amd._capi import NR_QP
xs_b = torch.randn(ntok, NR_KP, device=dev).mul_(0.3).to(torch.bfloat16); xs_b[:, 300] = 1.0; xs_b[:, 301:] = 0
dpre_b = torch.randn(ntok, NR_QP, device=dev).mul_(0.1).to(torch.bfloat16)
kern = {
  'pack_qkv32': lambda: ck(lib.nr_pack_qkv32(*wargs, Wp32.data_ptr(), bp32.data_ptr(), st())),
  'proj_train(x_save,drop)': lambda: ck(lib.nr_qkv_proj_fwd(ids.data_ptr(), table.data_ptr(), V, Wp32.data_ptr(), bp32.data_ptr(), qkv.data_ptr(), xs.data_ptr(), T, 20, p, 1, st())),
  'proj(no x_save,drop)': lambda: ck(lib.nr_qkv_proj_fwd(ids.data_ptr(), table.data_ptr(), V, Wp32.data_ptr(), bp32.data_ptr(), qkv.data_ptr(), None, T, 20, p, 1, st())),
  'proj(no x_save,no drop)': lambda: ck(lib.nr_qkv_proj_fwd(ids.data_ptr(), table.data_ptr(), V, Wp32.data_ptr(), bp32.data_ptr(), qkv.data_ptr(), None, T, 20, 0.0, 0, st())),
  'attn_fwd(drop)': lambda: ck(lib.nr_attn_fwd(qkv.data_ptr(), ctx.data_ptr(), None, T, 20, p, 1, st())),
  'attn_pool_fwd(drop)': lambda: ck(lib.nr_attn_pool_fwd(qkv.data_ptr(), ctx.data_ptr(), None, Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), nv.data_ptr(), NR_D, awo.data_ptr(), T, 20, 20, p, 1, st())),
  'additive_fwd': lambda: ck(lib.nr_additive_fwd(ctx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), nv.data_ptr(), awo.data_ptr(), T, 20, st())),
  'attn_fwd(no drop)': lambda: ck(lib.nr_attn_fwd(qkv.data_ptr(), ctx.data_ptr(), None, T, 20, 0.0, 0, st())),
  'mhsa_fwd2_train(drop)': lambda: ck(lib.nr_mhsa_fwd_len(ids.data_ptr(), table.data_ptr(), V, None, Wp.data_ptr(), bp.data_ptr(), ctx.data_ptr(), qs.data_ptr(), ks.data_ptr(), vts.data_ptr(), xs.data_ptr(), None, T, 20, p, 1, st())),
  'mhsa_fwd2_infer': lambda: ck(lib.nr_mhsa_fwd(ids.data_ptr(), table.data_ptr(), V, None, Wp.data_ptr(), bp.data_ptr(), ctx.data_ptr(), None, None, None, T, 20, 0.0, 0, st())),
  'attn_bwd_rowmajor': lambda: ck(lib.nr_attn_bwd_len(qs.data_ptr(), ks.data_ptr(), vts.data_ptr(), dctx.data_ptr(), NR_KP, aw.data_ptr(), go.data_ptr(), dqkv.data_ptr(), None, T, 20, p, 1, st())),
  'dX_gemm_hand': lambda: ck(lib.nr_dx_gemm(dqkv_b.data_ptr(), WdX.data_ptr(), dX_hand.data_ptr(), ntok, st())),
  'dX_gemm_hipblaslt': lambda: torch.nn.functional.linear(dqkv_b, WpT),
  'attn_bwd_hm': lambda: ck(lib.nr_attn_bwd_hm(qkv.data_ptr(), dctx.data_ptr(), NR_KP, aw.data_ptr(), go.data_ptr(), dqkv.data_ptr(), None, T, 20, p, 1, st())),
}
only = os.environ.get('KB_ONLY')
res = {}
for name, fn in kern.items():
    if only and only not in name:
        continue
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    res[name] = e0.elapsed_time(e1) / n * 1e3
    print(f'{name:28s} {res[name]:10.1f} us', flush=True)
if 'dX_gemm_hand' in res and 'dX_gemm_hipblaslt' in res:
    ref = torch.nn.functional.linear(dqkv_b, WpT).float()
    err = (dX_hand.float() - ref).abs().max().item() / ref.abs().max().item()
    print(f'dX hand vs hipBLASLt: max rel diff {err:.3g}; TFLOP/s (algorithmic 2*900*300 per token): hand {ntok * 2 * 900 * 300 / res["dX_gemm_hand"] / 1e6:.0f}, '
          f'hipBLASLt {ntok * 2 * 900 * 300 / res["dX_gemm_hipblaslt"] / 1e6:.0f}')
flop_proj = ntok * 2 * 300 * 900
out = {'B': B, 'us': res, 'env': {k: v for k, v in os.environ.items() if k.startswith('NR_')}}
for k in res:
    if k.startswith('proj'):
        out.setdefault('proj_TFLOPs', {})[k] = flop_proj / res[k] / 1e6
print(json.dumps(out))

# KB_SWEEP=1: phase decomposition of the two forward kernels in this process (the debug switches are re-read by every call)
if os.environ.get('KB_SWEEP'):
    def _t(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    print('qkv_proj phases (NR_PROJ_DEBUG: 1 no table loads, 2 no MFMAs, 4 no Q/K/V stores, 8 no x_save stores, 16 no weight copies; 32 = debug build, nothing off)')
    for d in (32, 33, 34, 36, 40, 48, 44, 45, 47, 63):
        os.environ['NR_PROJ_DEBUG'] = str(d)
        print(f'  NR_PROJ_DEBUG={d:2d}: {_t(kern["proj_train(x_save,drop)"]):8.1f} us', flush=True)
    os.environ.pop('NR_PROJ_DEBUG')
    print('attn_pool_fwd phases (NR_ATTNF_DEBUG: 1 no operand loads, 2 no exp / normalisation, 4 no ctx stores; 8 = debug build, nothing off)')
    for d in (8, 9, 10, 12, 13, 15):
        os.environ['NR_ATTNF_DEBUG'] = str(d)
        print(f'  NR_ATTNF_DEBUG={d:2d}: pooled {_t(kern["attn_pool_fwd(drop)"]):8.1f} us   plain {_t(kern["attn_fwd(drop)"]):8.1f} us', flush=True)
    os.environ.pop('NR_ATTNF_DEBUG')
