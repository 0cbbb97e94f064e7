"""The general ring GEMMs (csrc/k_gemm.h) on MI355X at the product's shapes, against numpy on the bf16 operands (float64 accumulation):
LSTUR's x W_ih^T / dGi W_ih (NT), dGi^T X / dGh^T H (TN, 256 x 256 tiles), the conv encoders' 3-tap weight gradient (TN, 320 x 256 tiles,
virtual operand rows) and the NRMS projection gradient (TN, 256 x 320 tiles)."""
import pytest
from tests import kernel_checks_gemm as kg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from tests.backends import GpuBackend
    return GpuBackend()


def test_gemm_nt_gru_input_projection(be): kg.check_gemm_nt(be, M=5131, N=2736, K=928)                 # x W_ih^T: 11 column tiles, the last one partial
def test_gemm_nt_gru_dx(be): kg.check_gemm_nt(be, M=4099, N=900, K=2752, lda=2752, ldb=2752, ldc=900)   # dGi W_ih: 86 chunks, 900 = 3.5 column tiles
def test_gemm_nt_small_and_strided(be): kg.check_gemm_nt(be, M=70, N=40, K=32, lda=40, ldb=64, ldc=44); kg.check_gemm_nt(be, M=300, N=290, K=96)
def test_gemm_tn_gru_weight_gradient(be): kg.check_gemm_tn(be, n_tok=25600, M=2752, ldg=2752, ncol=928, ldx=928)
def test_gemm_tn_conv_three_taps(be): kg.check_gemm_tn(be, n_tok=40003, M=320, ldg=320, ncol=320, ldx=320, taps=3)
def test_gemm_tn_projection_gradient(be): kg.check_gemm_tn(be, n_tok=30011, M=960, ldg=960, ncol=320, ldx=320)
def test_gemm_tn_ragged_partitions(be): kg.check_gemm_tn(be, n_tok=70, M=40, ncol=24, ldx=24, P=8); kg.check_gemm_tn(be, n_tok=300, M=330, ldg=336, ncol=200, ldx=208)
def test_transpose(be): kg.check_transpose(be); kg.check_transpose(be, R=2736, C=928, lds=928, ldd=2752)
