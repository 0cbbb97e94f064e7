"""Backend-agnostic parity checks of the general ring GEMMs (csrc/k_gemm.h): nr_gemm_nt, nr_gemm_tn (plain and 3-tap virtual operand),
nr_transpose_bf16, nr_sum_parts against numpy (bf16 operands as the kernel sees them, float64 accumulation)."""
import numpy as np

from tests.backends import bf16_to_f32, f32_to_bf16
from tests import kernel_checks as kc


def check_gemm_nt(be, M=300, N=290, K=96, lda=None, ldb=None, ldc=None, seed=1):
    """C[m][n] = sum_k A[m][k] B[n][k]; rows / columns beyond M / N of the last tiles are never stored; padding of C's rows untouched."""
    lda = K if lda is None else lda
    ldb = K if ldb is None else ldb
    ldc = N if ldc is None else ldc
    rng = np.random.default_rng(seed)
    A = f32_to_bf16(rng.normal(0, 0.5, size=(M, lda)).astype(np.float32))
    B = f32_to_bf16(rng.normal(0, 0.5, size=(N, ldb)).astype(np.float32))
    C = be.poison((M, ldc), np.float32)
    kc.ck(be, be.lib.nr_gemm_nt(be.ptr(be.dev(A)), lda, be.ptr(be.dev(B)), ldb, be.ptr(C), ldc, M, N, K, be.stream))
    be.sync()
    got = be.np(C)
    ref = bf16_to_f32(A[:, :K]).astype(np.float64) @ bf16_to_f32(B[:, :K]).astype(np.float64).T
    np.testing.assert_allclose(got[:, :N], ref, rtol=0, atol=2e-5 * np.sqrt(K) + 1e-6)
    if ldc > N:
        assert np.isnan(got[:, N:]).all(), 'columns >= N of C must not be written'
    z = be.empty((64,), np.float32)
    assert be.lib.nr_gemm_nt(None, lda, be.ptr(z), ldb, be.ptr(C), ldc, M, N, K, be.stream) != 0 and b'nr_gemm_nt' in be.lib.nr_last_error()
    assert be.lib.nr_gemm_nt(be.ptr(z), lda, be.ptr(z), ldb, be.ptr(C), ldc, M, N, 40, be.stream) != 0        # K not a multiple of 32
    assert be.lib.nr_gemm_nt(be.ptr(z), lda, be.ptr(z), ldb, be.ptr(C), ldc, 0, N, K, be.stream) == 0         # empty: nothing launched


def check_gemm_tn(be, n_tok=300, M=290, ldg=None, ncol=200, ldx=None, taps=1, seed=2, P=None):
    """out[p][m][n] = sum over partition p's tokens of G[tok][m] X[tok + n // tapw][n % tapw]; the partitions sum to G^T X (taps = 1) or to the
    three tap products side by side (taps = 3: X has n_tok + 2 rows); every partition is written (empty ones as zeros)."""
    ldg = (M + 7) // 8 * 8 if ldg is None else ldg
    ldx = ncol if ldx is None else ldx
    rng = np.random.default_rng(seed)
    G = f32_to_bf16(rng.normal(0, 0.5, size=(n_tok, ldg)).astype(np.float32))
    X = f32_to_bf16(rng.normal(0, 0.5, size=(n_tok + taps - 1, ldx)).astype(np.float32))
    N = taps * ncol
    P = be.lib.nr_gemm_tn_parts(M, N, n_tok) if P is None else P
    assert P > 0 and P % 8 == 0
    out = be.poison((P, M, N), np.float32)
    zeros = be.empty((64,), np.uint16)
    kc.ck(be, be.lib.nr_gemm_tn(be.ptr(be.dev(G)), ldg, M, be.ptr(be.dev(X)), ldx, ncol, taps, be.ptr(zeros), be.ptr(out), N, n_tok, P, be.stream))
    be.sync()
    got = be.np(out).astype(np.float64)
    assert np.isfinite(got).all()
    Gf, Xf = bf16_to_f32(G[:, :M]).astype(np.float64), bf16_to_f32(X).astype(np.float64)
    ref = np.concatenate([Gf.T @ Xf[t:t + n_tok, :ncol] for t in range(taps)], axis=1)
    np.testing.assert_allclose(got.sum(0), ref, rtol=0, atol=2e-5 * np.sqrt(n_tok) + 1e-6)
    tpp = ((n_tok + P - 1) // P + 31) // 32 * 32
    hi = min(tpp, n_tok)
    ref0 = np.concatenate([Gf[:hi].T @ Xf[t:t + hi, :ncol] for t in range(taps)], axis=1)
    np.testing.assert_allclose(got[0], ref0, rtol=0, atol=2e-5 * np.sqrt(hi) + 1e-6)
    if (P - 1) * tpp >= n_tok:
        assert not got[P - 1].any(), 'an empty partition must be written as zeros'
    # nr_sum_parts: the partitions summed in a fixed order, optionally accumulated into the destination
    n = M * N
    if n % 4 == 0:
        dst = be.dev(np.full((M, N), 0.25, dtype=np.float32))
        kc.ck(be, be.lib.nr_sum_parts(be.ptr(out), P, n, be.ptr(dst), 1, be.stream))
        be.sync()
        np.testing.assert_allclose(be.np(dst), 0.25 + be.np(out).astype(np.float64).sum(0), rtol=0, atol=1e-4)
        kc.ck(be, be.lib.nr_sum_parts(be.ptr(out), P, n, be.ptr(dst), 0, be.stream))
        be.sync()
        np.testing.assert_allclose(be.np(dst), be.np(out).astype(np.float64).sum(0), rtol=0, atol=1e-4)
    assert be.lib.nr_gemm_tn(None, ldg, M, be.ptr(zeros), ldx, ncol, taps, be.ptr(zeros), be.ptr(out), N, n_tok, P, be.stream) != 0
    assert b'nr_gemm_tn' in be.lib.nr_last_error()
    assert be.lib.nr_gemm_tn(be.ptr(zeros), ldg, M, be.ptr(zeros), ldx, ncol, taps, be.ptr(zeros), be.ptr(out), N, n_tok, 12, be.stream) != 0      # P % 8


def check_gemm_tn_scale(be, n_tok=1356803, M=320, ncol=320, taps=3, seed=12, chunk=65536):
    """nr_gemm_tn at the launch size of the bench (NAML's abstracts: 1,356,800 seqpad rows per step, three taps: src/model/NAML/news_encoder.py:21-37
    through autograd) against the float64 product accumulated over chunks of tokens; the sum over the partitions AND every single partition."""
    rng = np.random.default_rng(seed)
    G = f32_to_bf16(rng.normal(0, 0.5, size=(n_tok, M)).astype(np.float32))
    X = f32_to_bf16(rng.normal(0, 0.5, size=(n_tok + taps - 1, ncol)).astype(np.float32))
    N = taps * ncol
    P = be.lib.nr_gemm_tn_parts(M, N, n_tok)
    out = be.poison((P, M, N), np.float32)
    zeros = be.empty((64,), np.uint16)
    kc.ck(be, be.lib.nr_gemm_tn(be.ptr(be.dev(G)), M, M, be.ptr(be.dev(X)), ncol, ncol, taps, be.ptr(zeros), be.ptr(out), N, n_tok, P, be.stream))
    be.sync()
    got = be.np(out).astype(np.float64)
    assert np.isfinite(got).all()
    tpp = ((n_tok + P - 1) // P + 31) // 32 * 32                   # tokens per partition (the launcher's split)
    worst = 0.0
    for p_ in range(P):
        lo, hi = p_ * tpp, min((p_ + 1) * tpp, n_tok)
        ref = np.zeros((M, N))
        for c0 in range(lo, hi, chunk):
            c1 = min(c0 + chunk, hi)
            Gf = bf16_to_f32(G[c0:c1]).astype(np.float64)
            for t in range(taps):
                ref[:, t * ncol:(t + 1) * ncol] += Gf.T @ bf16_to_f32(X[c0 + t:c1 + t]).astype(np.float64)
        tol = 2e-5 * np.sqrt(max(hi - lo, 1)) + 1e-6
        err = np.abs(got[p_] - ref).max()
        assert err <= tol, f'partition {p_} (tokens {lo}..{hi}): max err {err:.3g} > {tol:.3g}'
        worst = max(worst, err / tol)
    return worst


def check_transpose(be, R=70, C=45, lds=48, ldd=72):
    rng = np.random.default_rng(3)
    src = rng.integers(0, 65536, size=(R, lds)).astype(np.uint16)
    dst = be.poison((C, ldd), np.uint16)
    kc.ck(be, be.lib.nr_transpose_bf16(be.ptr(be.dev(src)), R, C, lds, be.ptr(dst), ldd, be.stream))
    be.sync()
    got = be.np(dst)
    assert np.array_equal(got[:, :R], src[:, :C].T)
    assert (got[:, R:] == 0xFFFF).all(), 'padding columns of the destination must be left untouched'
