"""Kernel-logic parity on CPU: the HIP kernel sources compiled against the wave emulator (tests/emu) and
checked against the oracle.  Confirms tiling / fragment indexing / barriers before a GPU is spent on them;
the -m gpu twin (tests/test_kernels_gpu.py) runs the same checks on the product library."""
import pytest
from tests import kernel_checks as kc
from tests.backends import EmuBackend


@pytest.fixture(scope='module')
def be():
    return EmuBackend()


def test_probe_mfma(be): kc.check_probe_mfma(be)
def test_probe_tr16(be): kc.check_probe_tr16(be)
def test_gather(be): kc.check_gather(be, n_tokens=300)
def test_pack(be): kc.check_pack(be)
def test_mhsa_gather(be): kc.check_mhsa_gather(be, n_seq=6)
def test_mhsa_gather_dropout(be): kc.check_mhsa_gather(be, n_seq=5, p_drop=0.2)
def test_mhsa_dense(be): kc.check_mhsa_dense(be, n_seq=2)
def test_mhsa_x_save(be): kc.check_mhsa_x_save(be)
def test_mhsa_x_save_s50(be): kc.check_mhsa_x_save(be, S=50, n_seq=2, p_drop=0.0)
def test_mhsa_key_len_s20(be): kc.check_mhsa_key_len(be, S=20, n_seq=7)
def test_mhsa_key_len_s50(be): kc.check_mhsa_key_len(be, S=50, n_seq=3)
def test_attn_bwd_key_len_s20(be): kc.check_attn_bwd(be, S=20, n_seq=5, with_key_len=True)
def test_attn_bwd_key_len_s50(be): kc.check_attn_bwd(be, S=50, n_seq=2, with_key_len=True)
def test_additive_valid_s20(be): kc.check_additive_valid(be, S=20, n_seq=6, valid=13)
def test_additive_valid_s50(be): kc.check_additive_valid(be, S=50, n_seq=3, valid=31)
def test_additive_valid_s4(be): kc.check_additive_valid(be, S=4, n_seq=23, valid=2)
def test_additive_s20(be): kc.check_additive(be, S=20, n_seq=6)
def test_additive_s50(be): kc.check_additive(be, S=50, n_seq=3)
def test_score_dot(be): kc.check_score_dot(be)
def test_score_csr(be): kc.check_score_csr(be)
def test_impression_metrics(be): kc.check_impression_metrics(be)
def test_bad_args(be): kc.check_bad_args(be)
def test_attn_bwd_s20(be): kc.check_attn_bwd(be, S=20, n_seq=3)
def test_attn_bwd_s20_dctx_through_lds(be): kc.check_attn_bwd(be, S=20, n_seq=7, p_drop=0.2, ldc=320); kc.check_attn_bwd(be, S=20, n_seq=3, with_key_len=True, ldc=320)
def test_attn_bwd_s20_dropout(be): kc.check_attn_bwd(be, S=20, n_seq=2, p_drop=0.2)
def test_attn_bwd_s50(be): kc.check_attn_bwd(be, S=50, n_seq=1)
def test_additive_bwd_s20(be): kc.check_additive_bwd(be, S=20, n_seq=6)
def test_additive_bwd_scale_check_small(be): kc.check_additive_bwd_scale(be, S=20, n_seq=5, chunk=2)
def test_additive_bwd_s50(be): kc.check_additive_bwd(be, S=50, n_seq=3)
def test_additive_bwd_valid_length(be): kc.check_additive_bwd(be, S=20, n_seq=6, valid=13); kc.check_additive_bwd(be, S=50, n_seq=3, valid=37)
def test_additive_bwd_s50_register_resident():
    """k_pool2.h <50, 1, 4> (4 fifty-token sequences per workgroup; the default from 2048 sequences up, forced here with NR_POOL2_S50=2):
    full, partly filled and single workgroups."""
    import subprocess, sys, os
    env = dict(os.environ, NR_POOL2_S50='2')
    code = ("from tests.backends import EmuBackend; from tests import kernel_checks as k; be = EmuBackend(); "
            "assert be.lib.nr_additive_bwd_grid(9, 50) == 3; k.check_additive_bwd(be, S=50, n_seq=6); k.check_additive_bwd(be, S=50, n_seq=3); "
            "k.check_additive_bwd(be, S=50, n_seq=5, valid=37)")
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_additive_bwd_s50_lds_tile_variant():
    """NR_POOL2_S50=0: the LDS-tile backward for 50-token sequences (one sequence per workgroup), which the register-resident kernel of
    csrc/k_pool2.h replaced as the default."""
    import subprocess, sys, os
    env = dict(os.environ, NR_POOL2_S50='0')
    code = ("from tests.backends import EmuBackend; from tests import kernel_checks as k; be = EmuBackend(); "
            "assert be.lib.nr_additive_bwd_grid(9, 50) == 9; k.check_additive_bwd(be, S=50, n_seq=3)")
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_wgrad_unpack(be): kc.check_wgrad_unpack(be)
def test_wgrad_unpack_small_qdim(be): kc.check_wgrad_unpack(be, nc_w=1, nc_a=1, nwg=1, qdim=70)
def test_gather_bf16(be): kc.check_gather_bf16(be)
def test_scatter_add(be): kc.check_scatter_add(be)
def test_score_bwd(be): kc.check_score_bwd(be)
def test_score_ce(be): kc.check_score_ce(be)
def test_score_ce_target_scale_strided(be): kc.check_score_ce(be, B=9, C=5, with_target=True, with_scale=True, ld_extra=4)
def test_rows_to_f32(be): kc.check_rows_to_f32(be)
def test_rows_to_bf16(be): kc.check_rows_to_bf16(be)
def test_accum_many(be): kc.check_accum_many(be)
def test_accum_many_two_launches(be): kc.check_accum_many(be, n_items=53)
def test_scatter_sorted(be): kc.check_scatter_sorted(be)
def test_dropout_under_step_counter(be): kc.check_dropout_under_step_counter(be)
def test_scatter_sorted_nodrop(be): kc.check_scatter_sorted(be, n_tokens=130, V=9, p_drop=0.0)


def test_attn_bwd_persistent_loop(be, monkeypatch):
    """few workgroups, many (sequence, head) pairs per wave: exercises the pair loop and the register prefetch"""
    monkeypatch.setenv('NR_ATTN_BWD_MAX_WGS', '2')
    kc.check_attn_bwd(be, S=20, n_seq=4, p_drop=0.2)
    monkeypatch.setenv('NR_ATTN_BWD_MAX_WGS', '1')
    kc.check_attn_bwd(be, S=50, n_seq=1)


# ---- NAML / LSTUR kernels ------------------------------------------------------------------------------------------
from tests import kernel_checks_conv as kcc  # noqa: E402


def test_pack_conv(be): kcc.check_pack_conv(be)
def test_pack_conv_small(be): kcc.check_pack_conv(be, D=60, Fn=48)
def test_conv_fwd_s20(be): kcc.check_conv_fwd(be, S=20, n_seq=6)
def test_conv_fwd_s20_dropout(be): kcc.check_conv_fwd(be, S=20, n_seq=5, p_drop=0.2, tok_offset=140)
def test_conv_fwd_s50(be): kcc.check_conv_fwd(be, S=50, n_seq=3, p_drop=0.2)
def test_conv_fwd_valid_s20(be): kcc.check_conv_fwd_valid(be, S=20, n_seq=6, valid=13)
def test_conv_fwd_valid_s50(be): kcc.check_conv_fwd_valid(be, S=50, n_seq=3, valid=33)
def test_conv_fwd_gemm_s20(be): kcc.check_conv_fwd(be, S=20, n_seq=14, gemm=True)                      # the training forward as gather pass + persistent ring GEMM (csrc/k_convgemm.h, EPI): two tiles
def test_conv_fwd_gemm_s20_dropout(be): kcc.check_conv_fwd(be, S=20, n_seq=50, p_drop=0.2, tok_offset=140, gemm=True)      # 5 tiles over 3 "CUs": the stream crosses a tile boundary
def test_conv_fwd_gemm_s50(be): kcc.check_conv_fwd(be, S=50, n_seq=3, p_drop=0.2, gemm=True)
def test_conv_fwd_gemm_valid(be): kcc.check_conv_fwd_valid(be, S=20, n_seq=6, valid=13, gemm=True); kcc.check_conv_fwd_valid(be, S=50, n_seq=3, valid=33, gemm=True)
def test_conv_dgrad_s20(be): kcc.check_conv_dgrad(be, S=20, n_seq=5)
def test_conv_dgrad_s50(be): kcc.check_conv_dgrad(be, S=50, n_seq=3)
def test_conv_dgrad_gemm_form(be): kcc.check_conv_dgrad_gemm(be, S=20, n_seq=14); kcc.check_conv_dgrad_gemm(be, S=50, n_seq=3)       # 293 / 152 virtual rows: two tiles, the second partial
def test_conv_dgrad_gemm_tap_inner_order():
    """NR_CONVGEMM_PAIRS=0 (read once per process: a child): the tap-inner chunk order of the persistent conv data-gradient GEMM."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('from tests.backends import EmuBackend; from tests import kernel_checks_conv as kcc; be = EmuBackend(); '
            'kcc.check_conv_dgrad_gemm(be, S=20, n_seq=50); kcc.check_conv_dgrad_gemm(be, S=50, n_seq=3); print("tap-inner ok")')
    out = subprocess.run([sys.executable, '-c', code], cwd=root, env=dict(os.environ, NR_CONVGEMM_PAIRS='0'), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and 'tap-inner ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
def test_conv_dgrad_gemm_persistent_stream(be): kcc.check_conv_dgrad_gemm(be, S=20, n_seq=50)       # 1,049 virtual rows = 5 tiles over the emulator's 3 "CUs": two tiles per workgroup, the ring runs across the tile boundary
def test_conv_act_bwd(be): kcc.check_conv_act_bwd(be)
def test_additive_bwd_act_fused_s20(be): kcc.check_additive_bwd_act(be, S=20, n_seq=7); kcc.check_additive_bwd_act(be, S=20, n_seq=17)
def test_additive_bwd_act_two_kernels_s50(be): kcc.check_additive_bwd_act(be, S=50, n_seq=3)       # below 2048 sequences: LDS-tile kernel + conv_act_bwd


def test_additive_bwd_act_fused_s50():
    import subprocess, sys, os
    env = dict(os.environ, NR_POOL2_S50='2')        # the register-resident kernel for 50-token sequences regardless of the batch size
    code = ("from tests.backends import EmuBackend; from tests import kernel_checks_conv as k; be = EmuBackend(); "
            "k.check_additive_bwd_act(be, S=50, n_seq=6); k.check_additive_bwd_act(be, S=50, n_seq=3)")
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_additive_ex_s4(be): kcc.check_additive_ex(be, S=4, n_seq=23)
def test_additive_ex_s20(be): kcc.check_additive_ex(be, S=20, n_seq=5)
def test_additive_bwd_s4(be): kcc.check_additive_bwd_s4(be)
def test_element_tables(be): kcc.check_element_tables(be, ncat=37, dcat=20, T=50)
def test_row_scatters(be): kcc.check_row_scatters(be)


# ---- LSTUR GRU step kernels -----------------------------------------------------------------------------------------
from tests import kernel_checks_gru as kcg  # noqa: E402


def test_gru_ini(be): kcg.check_gru(be, B=5, N=4, Hd=900, I=900, lens=[4, 1, 3, 2, 4])
def test_gru_con_hidden_450(be): kcg.check_gru(be, B=3, N=3, Hd=450, I=900, seed=1)
def test_gru_two_batch_tiles(be): kcg.check_gru(be, B=19, N=2, Hd=48, I=40, seed=2)
def test_gru_rows_and_gate(be): kcg.check_gru_rows(be, B=9, N=4, R=7, Hd=48); kcg.check_gru_rows(be, B=5, N=3, R=4, Hd=450, seed=8)     # row-indexed inference sweep, gate kernel (evaluation); Hd % 4 != 0


def test_gru_lds_variant():
    """NR_GRU_LDS=1: the W_hh / W_hh^T tile staged in LDS and two sample tiles per wave (experimental knob) against the same oracle."""
    import subprocess, sys, os
    env = dict(os.environ, NR_GRU_LDS='1', NR_GRU_NB='2')
    code = ("from tests.backends import EmuBackend; from tests import kernel_checks_gru as k; be = EmuBackend(); assert be.lib.nr_gru_seq_buffers(37, 900, 3) == 2; "
            "k.check_gru(be, B=37, N=3, seed=4); k.check_gru(be, B=5, N=4, Hd=450, I=900, seed=5)")
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_dropout_mask_statistics(be):
    kc.check_dropout_mask_statistics(be)


# ---- flat pooling backward (csrc/k_pool3.h) -----------------------------------------------------------------------------
from tests import kernel_checks_pool3 as k3  # noqa: E402


def test_pool_flat_bad_args(be): k3.check_flat_bad_args(be)
def test_pool_flat_s20(be): k3.check_flat(be, S=20, n_seq=6)                      # 120 tokens: sequences straddle the 48-row groups
def test_pool_flat_s50(be): k3.check_flat(be, S=50, n_seq=5)
def test_pool_flat_s7(be): k3.check_flat(be, S=7, n_seq=30, seed=3)                  # 8 sequences in 48 tokens: the most one slot tile holds
def test_pool_flat_s4_views(be): k3.check_flat(be, S=4, n_seq=83, seed=6); k3.check_flat(be, S=5, n_seq=40, seed=7); k3.check_flat(be, S=6, n_seq=31, seed=8)     # 13 sequences per group (NAML's 4 views): the second slot tile
def test_pool_flat_valid_and_strided_y(be): k3.check_flat(be, S=20, n_seq=7, valid=13, y_stride=3 * 300)
def test_pool_flat_any_length_dpre_only(be): k3.check_flat(be, S=33, n_seq=4, with_dctx=False, seed=5)     # a length no forward kernel is instantiated for
def test_pool_flat_strided_g(be): k3.check_flat(be, S=20, n_seq=7, g_stride=900, seed=11); k3.check_flat_act(be, S=20, n_seq=9, g_stride=900, seed=12); k3.check_flat(be, S=4, n_seq=83, g_stride=304, seed=13)
def test_pool_flat_act_s20(be): k3.check_flat_act(be, S=20, n_seq=7)
def test_pool_flat_act_s50(be): k3.check_flat_act(be, S=50, n_seq=3); k3.check_flat_act(be, S=16, n_seq=10, seed=8)      # 16: four sequences in 48 tokens
def test_pool_flat_persistent_loop(be): k3.check_flat(be, S=20, n_seq=130, seed=9); k3.check_flat_act(be, S=50, n_seq=60, seed=4)    # > 24 groups: several iterations per wave
def test_pool_flat_group_boundaries(be):
    """Row counts around the 48-row group: one full group, one row more, fewer rows than a tile; sequences longer than a group."""
    k3.check_flat(be, S=48, n_seq=1, seed=1); k3.check_flat(be, S=49, n_seq=1, seed=2); k3.check_flat(be, S=7, n_seq=1, seed=3)
    k3.check_flat(be, S=200, n_seq=2, seed=4)
def test_pool_flat_act_lengths(be): k3.check_flat_act(be, S=16, n_seq=3, seed=5); k3.check_flat_act(be, S=17, n_seq=20, seed=6); k3.check_flat_act(be, S=128, n_seq=2, seed=7)


def test_scatter_sorted_long_runs_across_spans(be):
    """Zipf-like ids: runs far longer than a wave's span (256 positions) and than a sub-chunk (64), a span that starts inside the padding ids,
    spans shared by three runs -- the seams where a run continues in the neighbouring wave take the atomic path, everything else the direct one."""
    kc.check_scatter_sorted(be, n_tokens=1500, V=7, p_drop=0.2, seed=11)
    kc.check_scatter_sorted(be, n_tokens=700, V=3, p_drop=0.0, seed=12)


@pytest.mark.parametrize('S,n_seq,valid', [(20, 11, None), (50, 5, None), (20, 4, 13), (33, 3, None), (64, 2, 40)])
def test_additive_forward_whole_sequences_per_wave(be, S, n_seq, valid): kc.check_additive_flat(be, S=S, n_seq=n_seq, valid=valid)
