"""Per-kernel micro-benchmark on cuda:0 (HIP events on the launch stream).  Usage: python tools/kbench.py [B]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from news_recommendation_amd import _capi
from news_recommendation_amd._capi import NR_D, NR_KP, NR_NP, NR_QP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
V = 70976
dev = torch.device('cuda:0')
lib = _capi.load()
st = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device='cpu').manual_seed(0)
T = B * 53
table = torch.randn(V, NR_D, generator=g).to(dev)
# Zipf-ish token ids
rng = np.random.default_rng(0)
ids_np = np.minimum(rng.zipf(1.2, size=(T, 20)), V - 1).astype(np.int64)
ids = torch.from_numpy(ids_np).to(dev)
ids_u = torch.from_numpy(rng.integers(1, V, size=(T, 20)).astype(np.int64)).to(dev)
W = [torch.randn(300, 300, generator=g).mul_(0.05).to(dev) for _ in range(3)]
bb = [torch.randn(300, generator=g).mul_(0.05).to(dev) for _ in range(3)]
Wa = torch.randn(200, 300, generator=g).mul_(0.05).to(dev); ba = torch.zeros(200, device=dev); qv = torch.randn(200, generator=g).mul_(0.1).to(dev)
Wp = torch.empty(3 * NR_NP, NR_KP, dtype=torch.int16, device=dev); bp = torch.empty(3 * NR_NP, device=dev)
Wap = torch.empty(NR_QP, NR_KP, dtype=torch.int16, device=dev); bap = torch.empty(NR_QP, device=dev); qvp = torch.empty(NR_QP, device=dev)
ctx = torch.empty(T * 20, NR_KP, dtype=torch.int16, device=dev)
nv = torch.empty(T, NR_D, device=dev); aw = torch.empty(T, 20, device=dev)
ctxu = torch.empty(B * 50, NR_KP, dtype=torch.int16, device=dev)
uv = torch.empty(B, NR_D, device=dev); awu = torch.empty(B, 50, device=dev)
logits = torch.empty(B, 3, device=dev)
gout = torch.empty(T * 20, NR_D, device=dev)
ck = lambda rc: _capi.check(lib, rc)

def pack():
    ck(lib.nr_pack_qkv(W[0].data_ptr(), bb[0].data_ptr(), W[1].data_ptr(), bb[1].data_ptr(), W[2].data_ptr(), bb[2].data_ptr(), Wp.data_ptr(), bp.data_ptr(), st()))
    ck(lib.nr_pack_additive(Wa.data_ptr(), ba.data_ptr(), qv.data_ptr(), 200, Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), st()))

kern = {
  'pack': pack,
  'gather_zipf': lambda: ck(lib.nr_gather_rows_f32(ids.data_ptr(), table.data_ptr(), gout.data_ptr(), T * 20, NR_D, V, st())),
  'gather_uniform': lambda: ck(lib.nr_gather_rows_f32(ids_u.data_ptr(), table.data_ptr(), gout.data_ptr(), T * 20, NR_D, V, st())),
  'mhsa_news': lambda: ck(lib.nr_mhsa_fwd(ids.data_ptr(), table.data_ptr(), V, None, Wp.data_ptr(), bp.data_ptr(), ctx.data_ptr(), None, None, None, T, 20, 0.0, 0, st())),
  'mhsa_news_drop': lambda: ck(lib.nr_mhsa_fwd(ids.data_ptr(), table.data_ptr(), V, None, Wp.data_ptr(), bp.data_ptr(), ctx.data_ptr(), None, None, None, T, 20, 0.2, 1, st())),
  'additive_news': lambda: ck(lib.nr_additive_fwd(ctx.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), nv.data_ptr(), aw.data_ptr(), T, 20, st())),
  'mhsa_user': lambda: ck(lib.nr_mhsa_fwd(None, None, 0, nv.data_ptr() + 3 * B * NR_D * 4, Wp.data_ptr(), bp.data_ptr(), ctxu.data_ptr(), None, None, None, B, 50, 0.0, 0, st())),
  'additive_user': lambda: ck(lib.nr_additive_fwd(ctxu.data_ptr(), Wap.data_ptr(), bap.data_ptr(), qvp.data_ptr(), uv.data_ptr(), awu.data_ptr(), B, 50, st())),
  'score': lambda: ck(lib.nr_score_dot(nv.data_ptr(), uv.data_ptr(), logits.data_ptr(), B, 3, NR_D, st())),
}
res = {}
for name, fn in kern.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    res[name] = e0.elapsed_time(e1) / n * 1e3
    print(f'{name:18s} {res[name]:10.1f} us', flush=True)
tok = T * 20
print(json.dumps({'B': B, 'us': res,
                  'gather_GBps_zipf': tok * 1200 / res['gather_zipf'] / 1e3, 'gather_GBps_uniform': tok * 1200 / res['gather_uniform'] / 1e3,
                  'mhsa_news_TFLOPs': tok * (900 * 300 * 2 + 2 * 15 * 20 * 20 * 2) / res['mhsa_news'] / 1e6,
                  'fwd_total_us': sum(res[k] for k in ('pack', 'mhsa_news', 'additive_news', 'mhsa_user', 'additive_user', 'score'))}))
