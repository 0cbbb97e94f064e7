"""GPU parity tests proper: the product library (libnr_engine.so, hand-written HIP for gfx950) through the
C-ABI on cuda:0, checked against the oracle.  Same checks as tests/test_kernels_emu.py at larger sizes."""
import pytest
from tests import kernel_checks as kc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from tests.backends import GpuBackend
    return GpuBackend()


def test_probe_mfma(be): kc.check_probe_mfma(be)
def test_probe_tr16(be): kc.check_probe_tr16(be)
def test_gather(be): kc.check_gather(be, n_tokens=100003, V=70976)
def test_pack(be): kc.check_pack(be)
def test_mhsa_gather(be): kc.check_mhsa_gather(be, n_seq=1027, V=5000)
def test_mhsa_gather_small(be): kc.check_mhsa_gather(be, n_seq=3, V=300)
def test_mhsa_gather_dropout(be): kc.check_mhsa_gather(be, n_seq=515, V=5000, p_drop=0.2)
def test_mhsa_dense(be): kc.check_mhsa_dense(be, n_seq=131)
def test_mhsa_x_save(be): kc.check_mhsa_x_save(be, n_seq=1027, V=5000)
def test_mhsa_x_save_s50(be): kc.check_mhsa_x_save(be, S=50, n_seq=33, p_drop=0.0)
def test_mhsa_key_len_s20(be): kc.check_mhsa_key_len(be, S=20, n_seq=1027)
def test_mhsa_key_len_s50(be): kc.check_mhsa_key_len(be, S=50, n_seq=131)
def test_attn_bwd_key_len_s20(be): kc.check_attn_bwd(be, S=20, n_seq=203, with_key_len=True)
def test_attn_bwd_key_len_s50(be): kc.check_attn_bwd(be, S=50, n_seq=37, with_key_len=True)
def test_additive_valid_s20(be): kc.check_additive_valid(be, S=20, n_seq=1027, valid=13)
def test_additive_valid_s50(be): kc.check_additive_valid(be, S=50, n_seq=131, valid=31)
def test_additive_valid_s4(be): kc.check_additive_valid(be, S=4, n_seq=2047, valid=3)
def test_additive_s20(be): kc.check_additive(be, S=20, n_seq=1027)
def test_additive_s50(be): kc.check_additive(be, S=50, n_seq=131)
def test_score_dot(be): kc.check_score_dot(be, B=513, C=3)
def test_score_csr(be): kc.check_score_csr(be, n_news=5000, n_users=300, n_impr=1000)
def test_impression_metrics(be): kc.check_impression_metrics(be, n_impr=3001)
def test_bad_args(be): kc.check_bad_args(be)
def test_attn_bwd_s20(be): kc.check_attn_bwd(be, S=20, n_seq=203)
def test_attn_bwd_s20_dctx_through_lds(be): kc.check_attn_bwd(be, S=20, n_seq=2500, p_drop=0.2, ldc=320); kc.check_attn_bwd(be, S=20, n_seq=203, with_key_len=True, ldc=320)
def test_attn_bwd_s20_dropout(be): kc.check_attn_bwd(be, S=20, n_seq=57, p_drop=0.2)
def test_attn_bwd_s50(be): kc.check_attn_bwd(be, S=50, n_seq=37)
def test_additive_bwd_s20(be): kc.check_additive_bwd(be, S=20, n_seq=1027)
def test_additive_bwd_s50(be): kc.check_additive_bwd(be, S=50, n_seq=131)
def test_additive_bwd_at_bench_scale(be): kc.check_additive_bwd_scale(be, S=20, n_seq=27136); kc.check_additive_bwd_scale(be, S=50, n_seq=27136 // 2 + 3)     # NRMS titles / NAML abstracts launch sizes vs the numpy oracle, chunked
def test_additive_bwd_valid_length(be): kc.check_additive_bwd(be, S=20, n_seq=1027, valid=13); kc.check_additive_bwd(be, S=50, n_seq=131, valid=37); kc.check_additive_bwd(be, S=50, n_seq=2051, valid=41)
def test_additive_bwd_s50_register_resident(be): kc.check_additive_bwd(be, S=50, n_seq=2051)      # k_pool2.h <50, 1, 4> (the default from 2048 sequences up): 4 per workgroup, the last one partly filled


def test_additive_bwd_s50_lds_tile_variant():
    """NR_POOL2_S50=0: the LDS-tile backward for 50-token sequences (one sequence per workgroup), which the register-resident kernel of
    csrc/k_pool2.h replaced as the default."""
    import subprocess, sys, os
    env = dict(os.environ, NR_POOL2_S50='0')
    code = ("from tests.backends import GpuBackend; from tests import kernel_checks as k; be = GpuBackend(); "
            "assert be.lib.nr_additive_bwd_grid(9, 50) == 9; k.check_additive_bwd(be, S=50, n_seq=131)")
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_wgrad_unpack(be): kc.check_wgrad_unpack(be, nc_w=32, nc_a=64, nwg=1696)
def test_wgrad_unpack_small_qdim(be): kc.check_wgrad_unpack(be, nc_w=1, nc_a=1, nwg=1, qdim=70)
def test_gather_bf16(be): kc.check_gather_bf16(be, n_tokens=100003, V=5000)
def test_scatter_add(be): kc.check_scatter_add(be, n_tokens=200001, V=3000)
def test_score_bwd(be): kc.check_score_bwd(be, B=513)
def test_score_ce(be): kc.check_score_ce(be, B=513, C=3)
def test_score_ce_target_scale_strided(be): kc.check_score_ce(be, B=130, C=5, with_target=True, with_scale=True, ld_extra=600)
def test_score_ce_64(be): kc.check_score_ce(be, B=7, C=64, with_target=True)
def test_rows_to_f32(be): kc.check_rows_to_f32(be, n=25601)
def test_rows_to_bf16(be): kc.check_rows_to_bf16(be, n=2561)
def test_accum_many(be): kc.check_accum_many(be, n_items=21)
def test_accum_many_two_launches(be): kc.check_accum_many(be, n_items=53)
def test_scatter_sorted(be): kc.check_scatter_sorted(be, n_tokens=200001, V=3000)
def test_dropout_under_step_counter(be): kc.check_dropout_under_step_counter(be)
def test_scatter_sorted_nodrop(be): kc.check_scatter_sorted(be, n_tokens=54321, V=70976, p_drop=0.0)


def test_attn_bwd_persistent_loop(be, monkeypatch):
    monkeypatch.setenv('NR_ATTN_BWD_MAX_WGS', '3')
    kc.check_attn_bwd(be, S=20, n_seq=40, p_drop=0.2)
    kc.check_attn_bwd(be, S=50, n_seq=9)
    monkeypatch.delenv('NR_ATTN_BWD_MAX_WGS')
    kc.check_attn_bwd(be, S=20, n_seq=2500)          # > 1536 workgroups of pairs: the production grid loops too


# ---- NAML / LSTUR kernels ------------------------------------------------------------------------------------------
from tests import kernel_checks_conv as kcc  # noqa: E402


def test_pack_conv(be): kcc.check_pack_conv(be)
def test_conv_fwd_s20(be): kcc.check_conv_fwd(be, S=20, n_seq=1027, V=5000)
def test_conv_fwd_s20_dropout(be): kcc.check_conv_fwd(be, S=20, n_seq=515, V=5000, p_drop=0.2, tok_offset=140)
def test_conv_fwd_s50(be): kcc.check_conv_fwd(be, S=50, n_seq=203, V=5000, p_drop=0.2)
def test_conv_fwd_valid_s20(be): kcc.check_conv_fwd_valid(be, S=20, n_seq=515, valid=13)
def test_conv_fwd_valid_s50(be): kcc.check_conv_fwd_valid(be, S=50, n_seq=131, valid=33)
def test_conv_fwd_gemm(be):
    """The training forward as gather pass + persistent ring GEMM (nr_conv3_fwd_gemm, csrc/k_convgemm.h EPI); 3,300 titles = 271 tiles: the stream
    crosses a tile boundary on some CUs."""
    kcc.check_conv_fwd(be, S=20, n_seq=1027, V=5000, gemm=True)
    kcc.check_conv_fwd(be, S=20, n_seq=3300, V=5000, p_drop=0.2, tok_offset=140, gemm=True)
    kcc.check_conv_fwd(be, S=50, n_seq=203, V=5000, p_drop=0.2, gemm=True)
    kcc.check_conv_fwd_valid(be, S=20, n_seq=515, valid=13, gemm=True)
    kcc.check_conv_fwd_valid(be, S=50, n_seq=131, valid=33, gemm=True)
def test_conv_dgrad_s20(be): kcc.check_conv_dgrad(be, S=20, n_seq=515)
def test_conv_dgrad_s50(be): kcc.check_conv_dgrad(be, S=50, n_seq=131)
def test_conv_dgrad_gemm_form(be): kcc.check_conv_dgrad_gemm(be, S=20, n_seq=2051); kcc.check_conv_dgrad_gemm(be, S=50, n_seq=1031)
def test_conv_dgrad_gemm_many_tiles_per_cu(be): kcc.check_conv_dgrad_gemm(be, S=20, n_seq=7013)       # 575 tiles: 2-3 per CU, the ring runs across tile boundaries
def test_conv_act_bwd(be): kcc.check_conv_act_bwd(be, S=20, n_seq=1027)
def test_additive_bwd_act_fused_s20(be): kcc.check_additive_bwd_act(be, S=20, n_seq=1027)
def test_additive_bwd_act_fused_s50(be): kcc.check_additive_bwd_act(be, S=50, n_seq=2051)
def test_additive_bwd_act_two_kernels_s50(be): kcc.check_additive_bwd_act(be, S=50, n_seq=131)
def test_additive_ex_s4(be): kcc.check_additive_ex(be, S=4, n_seq=2047)
def test_additive_ex_s50(be): kcc.check_additive_ex(be, S=50, n_seq=131)
def test_additive_bwd_s4(be): kcc.check_additive_bwd_s4(be, n_seq=2047)
def test_element_tables(be): kcc.check_element_tables(be)
def test_row_scatters(be): kcc.check_row_scatters(be, n=54321, rows=275)


# ---- LSTUR GRU step kernels -----------------------------------------------------------------------------------------
from tests import kernel_checks_gru as kcg  # noqa: E402


def test_gru_ini(be): kcg.check_gru(be, B=133, N=50, Hd=900, I=900)
def test_gru_con_hidden_450(be): kcg.check_gru(be, B=70, N=20, Hd=450, I=900, seed=1)
def test_gru_rows_and_gate(be): kcg.check_gru_rows(be, B=300, N=50, R=500, Hd=900); kcg.check_gru_rows(be, B=33, N=7, R=20, Hd=450, seed=5)


def test_gru_lds_default_large_batch(be):
    """From 256 samples up the default step kernels stage the W_hh / W_hh^T tile in LDS (two sample tiles per wave in the forward)."""
    kcg.check_gru(be, B=300, N=12, Hd=900, I=900, seed=3)
    kcg.check_gru(be, B=270, N=8, Hd=450, I=900, seed=4)


def test_gru_full_grid_batch_512(be):
    """B = 512 (57 unit tiles x 4 sample groups = 228 workgroups, one per CU): the whole-sweep entry points nr_gru_fwd_seq_n / nr_gru_bwd_seq_n
    against the per-step launches, bit for bit, with ragged lengths."""
    kcg.check_gru(be, B=512, N=12, Hd=900, I=900, seed=6)
    kcg.check_gru(be, B=512, N=9, Hd=450, I=900, seed=7, lens=[1 + (7 * i) % 9 for i in range(512)])


def test_gru_persistent_sweeps_and_step_form(be):
    """The sweep entry points take the persistent XCD-local form (csrc/k_gru_persist.h) on a 256-CU device; check_gru compares the forward sweep bit
    for bit with the step launches and the backward sweep (another summation order) to rounding, and both with the float64 oracle.  Every
    setting of the knob (read per call: 3 both sweeps, 1 forward only, 2 backward only, 0 step launches), the error words clean afterwards, and
    batches that leave some XCDs without samples (B = 40: three sample tiles for eight XCDs) or with an odd number of sample tiles."""
    import os
    from news_recommendation_amd import ops_gru
    old = os.environ.get('NR_GRU_PERSIST')
    try:
        for knob in ('3', '1', '2', '0'):
            os.environ['NR_GRU_PERSIST'] = knob
            kcg.check_gru(be, B=512, N=7, Hd=900, I=900, seed=8, lens=[1 + (5 * i) % 7 for i in range(512)])
            kcg.check_gru(be, B=40, N=5, Hd=900, I=900, seed=9)
            kcg.check_gru(be, B=129, N=4, Hd=450, I=900, seed=10)
            assert ops_gru.persist_status() == (0, 0)
    finally:
        if old is None:
            os.environ.pop('NR_GRU_PERSIST', None)
        else:
            os.environ['NR_GRU_PERSIST'] = old


def test_gru_register_only_variant():
    """NR_GRU_LDS=0 (the round-1 default: operands straight from L2 into registers) stays selectable; knobs are read once per process,
    hence the subprocess."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NR_GRU_LDS='0')
    code = ("from tests.backends import GpuBackend; from tests import kernel_checks_gru as k; be = GpuBackend(); "
            "k.check_gru(be, B=300, N=12, Hd=900, I=900, seed=3); k.check_gru(be, B=70, N=8, Hd=450, I=900, seed=4)")
    r = subprocess.run([sys.executable, '-c', code], env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]


def test_dropout_mask_statistics(be):
    """Keep rate and decorrelation of the counter-based dropout generator on the hardware build (same check as on the emulator)."""
    kc.check_dropout_mask_statistics(be)


def test_dropout_mask_bits_equal_cpu_emulation(be):
    """The masks the GPU kernels draw are the masks the CPU build of the same sources draws, bit for bit (an independent check of the
    hardware's integer path: mul_lo / shifts / compares of csrc/nr_common.h)."""
    import numpy as np
    from tests.backends import EmuBackend
    emu = EmuBackend()
    for p, seed, site in ((0.2, 777, 1), (0.2, 2 ** 40 + 12345, 2), (0.5, 3, 1)):
        a = kc.export_mask(be, 100003, p, seed, site)
        b = kc.export_mask(emu, 100003, p, seed, site)
        assert np.array_equal(a, b)


# ---- flat pooling backward (csrc/k_pool3.h): the default of every pooling level's backward ------------------------------------------
from tests import kernel_checks_pool3 as k3  # noqa: E402


def test_pool_flat_bad_args(be): k3.check_flat_bad_args(be)
def test_pool_flat_small(be): k3.check_flat(be, S=20, n_seq=6); k3.check_flat(be, S=50, n_seq=5); k3.check_flat(be, S=7, n_seq=30, seed=3)
def test_pool_flat_s4_views(be): k3.check_flat(be, S=4, n_seq=28160, seed=6); k3.check_flat(be, S=5, n_seq=4001, seed=7); k3.check_flat(be, S=6, n_seq=3001, seed=8)     # NAML's view level at the bench's launch size; the second slot tile
def test_pool_flat_valid_strided_any_length(be):
    k3.check_flat(be, S=20, n_seq=1027, valid=13, y_stride=3 * 300); k3.check_flat(be, S=50, n_seq=131, valid=37)
    k3.check_flat(be, S=33, n_seq=517, with_dctx=False, seed=5)
def test_pool_flat_strided_g(be): k3.check_flat(be, S=20, n_seq=1027, g_stride=900, seed=11); k3.check_flat_act(be, S=20, n_seq=2051, g_stride=900, seed=12); k3.check_flat(be, S=4, n_seq=4001, g_stride=304, seed=13)
def test_pool_flat_act(be): k3.check_flat_act(be, S=20, n_seq=1027); k3.check_flat_act(be, S=50, n_seq=2051); k3.check_flat_act(be, S=16, n_seq=333, seed=8)
def test_pool_flat_at_bench_scale(be): k3.check_flat_scale(be, S=20, n_seq=27136); k3.check_flat_scale(be, S=50, n_seq=27136 // 2 + 3)
def test_pool_flat_act_at_bench_scale(be): k3.check_flat_scale(be, S=20, n_seq=27136, act=True); k3.check_flat_scale(be, S=50, n_seq=27136 // 2 + 3, act=True)


@pytest.mark.parametrize('S,n_seq,valid', [(20, 5000, None), (50, 2100, None), (20, 999, 13), (50, 513, 37), (33, 700, None)])
def test_additive_forward_whole_sequences_per_wave(be, S, n_seq, valid):
    """csrc/k_pool4.h at launch sizes that give every workgroup several groups per wave and a ragged last group."""
    kc.check_additive_flat(be, S=S, n_seq=n_seq, valid=valid)
