"""The general ring GEMMs (csrc/k_gemm.h) on the CPU wave emulator: tile / fragment / swizzle / ring logic against numpy before a GPU is
spent on them (tests/test_gemm_gpu.py repeats the checks at the product's shapes)."""
import pytest
from tests import kernel_checks_gemm as kg
from tests.backends import EmuBackend


@pytest.fixture(scope='module')
def be():
    return EmuBackend()


def test_gemm_nt_small(be): kg.check_gemm_nt(be, M=300, N=290, K=96)
def test_gemm_nt_strides_and_one_chunk(be): kg.check_gemm_nt(be, M=70, N=40, K=32, lda=40, ldb=64, ldc=44)
def test_gemm_nt_many_chunks(be): kg.check_gemm_nt(be, M=33, N=257, K=320)
def test_gemm_tn_wide(be): kg.check_gemm_tn(be, n_tok=300, M=330, ldg=336, ncol=200, ldx=208)            # 256 x 256 tiles, two row tiles
def test_gemm_tn_320_rows(be): kg.check_gemm_tn(be, n_tok=200, M=320, ncol=288, ldx=320)                  # 320 x 256 tile, two column tiles
def test_gemm_tn_320_columns(be): kg.check_gemm_tn(be, n_tok=130, M=330, ldg=336, ncol=320, ldx=320)            # 256 x 320 tile (projection gradients), two row tiles
def test_gemm_tn_three_taps(be): kg.check_gemm_tn(be, n_tok=150, M=304, ldg=320, ncol=320, ldx=320, taps=3, P=8)      # virtual [x[t], x[t+1], x[t+2]] rows
def test_gemm_tn_ragged_partitions(be): kg.check_gemm_tn(be, n_tok=70, M=40, ncol=24, ldx=24, P=8)         # partitions of 32 tokens, the last ones empty
def test_transpose(be): kg.check_transpose(be)
