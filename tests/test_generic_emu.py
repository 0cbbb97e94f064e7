"""The general-geometry kernels (csrc/k_generic.h: config knobs away from 300 / 15 / 300 / 3) on the CPU wave emulator against float64
restatements of the reference's formulas; tests/test_generic_gpu.py repeats them on MI355X and drives the drop-in models on them."""
import pytest
from tests import kernel_checks_generic as kg
from tests.backends import EmuBackend


@pytest.fixture(scope='module')
def be():
    return EmuBackend()


def test_attention_core_with_key_lengths(be): kg.check_attn(be, n_seq=5, S=13, H=4, dk=10)
def test_attention_core_full_width(be): kg.check_attn(be, n_seq=2, S=64, H=2, dk=32, with_len=False)
def test_additive_pooling(be): kg.check_additive(be, n_seq=6, S=11, D=52, Q=24)
def test_additive_pooling_valid_prefix(be): kg.check_additive(be, n_seq=3, S=20, D=36, Q=17, valid=7)
def test_dropout_uses_the_exported_masks(be): kg.check_dropout(be)
@pytest.mark.parametrize('w', [1, 3, 5])
def test_convolution_any_window(be, w): kg.check_conv(be, n_seq=4, S=9, D=20, F=24, w=w)
def test_relu(be): kg.check_relu(be)
def test_split_operand_linear(be): kg.check_split_linear(be)
