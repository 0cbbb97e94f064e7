"""Split training forward (csrc/k_proj.h) on the real library (cuda:0): the checks of tests/test_proj_emu.py at MI355X-filling sizes, and the
autograd path of ops._EncoderFn with the split form against the register-resident form."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from tests.backends import GpuBackend
    return GpuBackend()


def test_pack32(be):
    from tests import kernel_checks_proj as kp
    kp.check_pack32(be)


def test_pack_encoder_one_launch(be):
    from tests import kernel_checks_proj as kp
    kp.check_pack_encoder(be)


def test_qkv_proj(be):
    from tests import kernel_checks_proj as kp
    kp.check_qkv_proj(be, n_seq=13)
    kp.check_qkv_proj(be, n_seq=1999, V=5000)            # 312 full workgroups + a partly filled one


def test_qkv_proj_dropout(be):
    from tests import kernel_checks_proj as kp
    kp.check_qkv_proj(be, n_seq=257, V=900, p_drop=0.2)


def test_proj_attn(be):
    from tests import kernel_checks_proj as kp
    kp.check_proj_attn(be, n_seq=9)
    kp.check_proj_attn(be, n_seq=1031, V=3000)
    kp.check_proj_attn(be, n_seq=130, p_drop=0.2)
    kp.check_proj_attn(be, n_seq=67, with_key_len=True)


def test_attn_fwd_matches_fused(be):
    from tests import kernel_checks_proj as kp
    kp.check_attn_fwd_matches_fused(be, n_seq=403, V=2000)


def test_attn_pool(be):
    from tests import kernel_checks_proj as kp
    kp.check_attn_pool(be, n_seq=9)
    kp.check_attn_pool(be, n_seq=1)
    kp.check_attn_pool(be, n_seq=2050, V=2000, p_drop=0.2, seed=11, with_key_len=True)
    kp.check_attn_pool(be, n_seq=402, valid=13)


def test_attn_bwd_hm(be):
    from tests import kernel_checks_proj as kp
    kp.check_attn_bwd_hm(be, n_seq=5)
    kp.check_attn_bwd_hm(be, n_seq=700, p_drop=0.2, with_key_len=True)


def test_dx_gemm(be):
    from tests import kernel_checks_proj as kp
    kp.check_dx_gemm(be, n_tok=300)
    kp.check_dx_gemm(be, n_tok=20000 + 77, seed=32)


def test_dx_gemm_stream_form():
    """NR_DX_STREAM=1: nr_dx_gemm in the persistent stream kernel (csrc/k_convgemm.h, PLAIN) over its row-major operand (the switch is read
    once per process: a child process); 513 tiles = two or three per workgroup, the ring runs across tile boundaries."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('from tests.backends import GpuBackend; from tests import kernel_checks_proj as kp; be = GpuBackend(); '
            'kp.check_dx_gemm(be, n_tok=300); kp.check_dx_gemm(be, n_tok=20077, seed=32); kp.check_dx_gemm(be, n_tok=2 * 256 * 256 + 77, seed=34); '
            'kp.check_pack_encoder(be); print("stream ok")')
    out = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, NR_DX_STREAM='1'), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'stream ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_tn_gemm(be):
    from tests import kernel_checks_proj as kp
    kp.check_tn_gemm(be, n_tok=300, M=960, P=8)
    kp.check_tn_gemm(be, n_tok=77, M=208, P=8)
    kp.check_tn_gemm(be, n_tok=40000 + 60, M=960, seed=42)          # the library's partition count: 64 partitions x 8 slabs
    kp.check_tn_gemm(be, n_tok=40000 + 60, M=208, seed=43)


def test_proj_bad_args(be):
    from tests import kernel_checks_proj as kp
    kp.check_proj_bad_args(be)


def _encoder_run(split, seed=3):
    """loss + every gradient of one title-encoder call through ops.encode_titles in a child process (NR_FWD_SPLIT is read at import)."""
    code = f'''
import torch, numpy as np, sys
from news_recommendation_amd import ops
from news_recommendation_amd.dropin.model.NRMS.news_encoder import NewsEncoder
from news_recommendation_amd.default_config import NRMSConfig
torch.manual_seed({seed})
cfg = NRMSConfig
cfg.num_words = 3000
enc = NewsEncoder(cfg, None).to("cuda:0")
g = torch.Generator().manual_seed(5)
ids = torch.randint(0, 3000, (530, 20), generator=g).to("cuda:0")
ids[:, 14:] = 0
enc.eval()
out = enc({{"title": ids}})
w = torch.randn(out.shape, generator=g).to("cuda:0")
(out * w).sum().backward()
res = {{"out": out.detach().cpu().numpy()}}
for n, p_ in enc.named_parameters():
    res[n] = p_.grad.detach().cpu().numpy()
np.savez(sys.argv[1], **res)
'''
    import tempfile
    path = tempfile.mktemp(suffix='.npz')
    env = dict(os.environ, NR_FWD_SPLIT='1' if split else '0')
    r = subprocess.run([sys.executable, '-c', code, path], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    d = dict(np.load(path))
    os.unlink(path)
    return d


def test_encoder_autograd_split_vs_fused():
    """The title encoder's output and every parameter gradient with the split training forward == the register-resident form, to the
    accumulation-order noise of the projection (bf16 rounding of Q / K / V flips a last bit here and there): 1e-2 of each tensor's scale."""
    a, b = _encoder_run(True), _encoder_run(False)
    assert set(a) == set(b)
    # absolute floor for gradients that are analytically ~0 (W_K.bias: a per-query constant shift of the scores cancels in exp / (sum + 1e-8)):
    # 2e-2 of the largest bias gradient, as in tests/test_model_gpu.py
    floor = 2e-2 * max(np.abs(v).max() for k, v in b.items() if k.endswith('bias'))
    for k in a:
        scale = np.abs(b[k]).max() + 1e-30
        err = np.abs(a[k] - b[k]).max()
        assert err <= 1e-2 * scale + (floor if k.endswith('bias') else 0.0), f'{k}: max diff {err:.3g} on scale {scale:.3g}'


def test_attn_bwd_hm_vs_numpy_oracle_at_bench_scale(be):
    """The dominant kernel of the NRMS step at the launch size of BASELINE configs[1] (B = 512: 27,136 titles, 407,040 (title, head) pairs)
    against the numpy oracle directly; plus a dropout + key-length case whose n_seq leaves a partial last grid round."""
    from tests import kernel_checks_proj as kp
    kp.check_attn_bwd_hm_oracle(be, n_seq=27136)
    kp.check_attn_bwd_hm_oracle(be, n_seq=3001, p_drop=0.2, with_key_len=True)
