"""Split training forward (csrc/k_proj.h) on the CPU wave emulator: the same kernel sources, checked against the numpy oracle and against
the register-resident kernels.  The -m gpu twin is tests/test_proj_gpu.py."""
import pytest
from tests import kernel_checks_proj as kp
from tests.backends import EmuBackend


@pytest.fixture(scope='module')
def be():
    return EmuBackend()


def test_pack32(be): kp.check_pack32(be)
def test_pack_encoder_one_launch(be): kp.check_pack_encoder(be)
def test_qkv_proj(be): kp.check_qkv_proj(be, n_seq=13)          # 260 tokens: two full workgroups + a partly filled one
def test_qkv_proj_dropout(be): kp.check_qkv_proj(be, n_seq=7, p_drop=0.2)
def test_proj_attn(be): kp.check_proj_attn(be, n_seq=9)
def test_proj_attn_dropout(be): kp.check_proj_attn(be, n_seq=6, p_drop=0.2)
def test_proj_attn_key_len(be): kp.check_proj_attn(be, n_seq=7, with_key_len=True)
def test_attn_fwd_matches_fused(be): kp.check_attn_fwd_matches_fused(be, n_seq=6)
def test_attn_pool(be): kp.check_attn_pool(be, n_seq=9)                     # 2 full workgroups + one holding a single title
def test_attn_pool_dropout_key_len_valid(be): kp.check_attn_pool(be, n_seq=6, p_drop=0.2, seed=5, with_key_len=True, valid=17)
def test_attn_bwd_hm(be): kp.check_attn_bwd_hm(be, n_seq=5)
def test_attn_bwd_hm_dropout_key_len(be): kp.check_attn_bwd_hm(be, n_seq=4, p_drop=0.2, with_key_len=True)
def test_dx_gemm(be): kp.check_dx_gemm(be, n_tok=300)          # two tiles of 256 tokens, the second holding 44
def test_dx_gemm_stream_form():
    """NR_DX_STREAM=1 (read once per process: a child): the persistent stream kernel of csrc/k_convgemm.h (PLAIN) and its row-major operand;
    1,100 tokens = 5 tiles over the emulator's 3 "CUs": the ring runs across a tile boundary, partial last tile."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('from tests.backends import EmuBackend; from tests import kernel_checks_proj as kp; be = EmuBackend(); '
            'kp.check_dx_gemm(be, n_tok=300); kp.check_dx_gemm(be, n_tok=1100, seed=33); kp.check_pack_encoder(be); print("stream ok")')
    out = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, NR_DX_STREAM='1'), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and 'stream ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
def test_tn_gemm_dqkv(be): kp.check_tn_gemm(be, n_tok=300, M=960, P=8)          # 8 slabs x 8 partitions, ragged last chunk
def test_tn_gemm_dpre(be): kp.check_tn_gemm(be, n_tok=77, M=208, P=8)           # 2 slabs (128 + 80 rows), mostly empty partitions
def test_proj_bad_args(be): kp.check_proj_bad_args(be)
def test_attn_bwd_hm_vs_numpy_oracle(be):
    kp.check_attn_bwd_hm_oracle(be, n_seq=3)
    kp.check_attn_bwd_hm_oracle(be, n_seq=3, p_drop=0.2, with_key_len=True)
