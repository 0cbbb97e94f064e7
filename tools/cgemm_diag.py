"""Where does the persistent data-gradient GEMM (csrc/k_convgemm.h) differ from the LDS-tile kernel?  Rows with large errors mapped to (tile, workgroup, ordinal)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from news_recommendation_amd import _capi
from news_recommendation_amd._capi import NR_D, NR_KP
lib = _capi.load(); dev = torch.device('cuda:0'); st = lambda: torch.cuda.current_stream().cuda_stream
ck = lambda rc: _capi.check(lib, rc)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 7013
g = torch.Generator().manual_seed(0)
W = torch.randn(300, 1, 3, 300, generator=g).mul_(0.03).to(dev)
Wd2 = torch.empty(NR_KP, 3 * NR_KP, dtype=torch.int16, device=dev)
ck(lib.nr_pack_conv_dgrad(W.data_ptr(), 300, 300, Wd2.data_ptr(), st()))
Wc = torch.empty(3, NR_KP, NR_KP, dtype=torch.int16, device=dev); Wd = torch.empty_like(Wc); bc = torch.empty(NR_KP, device=dev)
b = torch.zeros(300, device=dev)
ck(lib.nr_pack_conv(W.data_ptr(), b.data_ptr(), 300, 300, Wc.data_ptr(), Wd.data_ptr(), bc.data_ptr(), st()))
dy = torch.zeros(n_seq * (S + 1) + 1, NR_KP)
tok = torch.randn(n_seq * S, 300, generator=g).mul_(0.1)
idx = torch.arange(n_seq * S)
dy[idx + idx // S + 1, :300] = tok
dyp = dy.to(torch.bfloat16).view(torch.int16).to(dev)
out_a = torch.full((n_seq * S, NR_KP), -1, dtype=torch.int16, device=dev)
out_b = torch.full((n_seq * S, NR_KP), -1, dtype=torch.int16, device=dev)
ck(lib.nr_conv3_dgrad_gemm(dyp.data_ptr(), Wd2.data_ptr(), out_a.data_ptr(), n_seq, S, st()))
ck(lib.nr_conv3_dgrad(dyp.data_ptr(), Wd.data_ptr(), out_b.data_ptr(), n_seq, S, st()))
torch.cuda.synchronize()
a = out_a.view(torch.bfloat16).float().cpu().numpy(); bb = out_b.view(torch.bfloat16).float().cpu().numpy()
unwritten = (out_a.cpu().numpy() == -1).all(axis=1)
err = np.abs(a[:, :300] - bb[:, :300]).max(axis=1)
bad = (err > 0.02 * np.abs(bb).max()) | unwritten | (a[:, 300:] != 0).any(axis=1)
print('tokens', n_seq * S, 'bad rows', int(bad.sum()), 'unwritten rows', int(unwritten.sum()), 'pad-nonzero rows', int((a[:, 300:] != 0).any(axis=1).sum()))
rows = np.nonzero(bad)[0]
sp = rows + rows // S + 1            # seqpad row of a token
vr = sp - 1
tiles = vr // 256
n_tiles = (n_seq * (S + 1) - 1 + 255) // 256
import collections
cnt = collections.Counter(tiles.tolist())
print('tiles', n_tiles, 'bad tiles', len(cnt))
cus = torch.cuda.get_device_properties(0).multi_processor_count
by_ord = collections.Counter((t // min(cus, n_tiles)) for t in cnt)
print('bad tiles by ordinal within their workgroup', dict(by_ord))
for t, c in sorted(cnt.items())[:12]:
    r = rows[tiles == t]
    loc = (vr[tiles == t] - t * 256)
    cols_bad = np.nonzero(np.abs(a[r[0], :320] - bb[r[0], :320]) > 0.02 * np.abs(bb).max())[0]
    print('tile', t, 'wg', t % min(cus, n_tiles), 'ord', t // min(cus, n_tiles), 'bad rows', c, 'local rows', loc[:6], '...', loc[-3:], 'bad cols of first', cols_bad[:8], len(cols_bad))
raw = out_a.cpu().numpy().view(np.uint16)
for r in rows[:6]:
    print('row', r, 'vals', a[r, 288:320], 'hex', [hex(x) for x in raw[r, 296:320]])
    print('   ref', bb[r, 288:300])
