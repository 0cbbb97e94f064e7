"""Autograd-capable host wrappers around the C-ABI kernels (libnr_engine.so).

PyTorch is plumbing here: it owns device memory, the current HIP stream and
autograd bookkeeping.  All encoder math runs in the hand-written HIP kernels;
the only library calls are the plain bf16 GEMMs of the backward pass
(``torch.mm`` -> hipBLASLt): dpre@Wa, dpre^T@ctx, dqkv@W, dqkv^T@X.

There is no CPU path: tensors must live on a ROCm device and the HIP library
must be built, otherwise a RuntimeError is raised.
"""
import collections
import os
import weakref

import torch

from . import _capi
from ._capi import NR_D, NR_KP, NR_NP, NR_QP, NR_HEADS, NR_DK, NR_LDG, NR_QKV_HM_SEQ, NR_K16

_BF16_AS_I16 = torch.int16   # bf16 buffers crossing the C-ABI are raw 16-bit words


def _lib():
    return _capi.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what} must be on the GPU: the NRMS engine has no CPU fallback "
                           f"(got device {t.device}); move the model with .to('cuda:0')")


def _ck(rc):
    _capi.check(_lib(), rc)


# ---- optional per-kernel timing with HIP events on the launch stream (used by bench.py) --------------------
_prof = None


class profile:
    """``with ops.profile() as rec:`` records a (start, end) HIP-event pair around every kernel launch / GEMM issued
    by this module on the current stream; ``rec.summary()`` -> {name: (n_launches, avg_us, total_us)}."""

    def __init__(self, only=None):
        self.events = {}
        self.only = only

    def __enter__(self):
        global _prof
        self._prev, _prof = _prof, self
        return self

    def __exit__(self, *a):
        global _prof
        _prof = self._prev

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for k, ev in self.events.items():
            us = [a.elapsed_time(b) * 1e3 for a, b in ev]
            out[k] = (len(us), sum(us) / len(us), sum(us))
        return out


def to_device_async(t, device):
    """Host -> device copy that does not drain the stream: the tensor is copied into a pinned staging buffer (rotating, reused only after
    the copy engine has read it) and crosses PCIe asynchronously on the copy stream.  A pageable source is staged by the runtime itself, and
    ``t.to(device)`` without ``non_blocking`` additionally ends in a stream synchronise."""
    if t.is_cuda or torch.device(device).type == 'cpu':
        return t
    r = _pinned.stage([t.detach().contiguous()], device, dim=0)
    return r[0][0]


class _PinnedRing:
    """Host staging for the DataLoader's per-position tensors (train.py:166-203 hands the model 1+K+N dicts of [B, ...] CPU tensors per
    step): they are written into ONE pinned buffer and cross PCIe as a single asynchronous copy (4.3 MB per NRMS step) instead of 53
    pageable ones, each of which would stall the stream.  Six buffers per (shape, dtype) rotate (two call sites may share a shape, and the
    host may run a step ahead of the GPU); a buffer is reused only after the copy
    that read it has completed (event)."""

    def __init__(self, depth=6):
        self.depth, self.slots = depth, {}

    def stage(self, parts, device, dim=1):
        """torch.stack(parts, dim) -> device tensor, via pinned memory when the parts live on the host."""
        p0 = parts[0]
        if p0.is_cuda:
            return torch.stack(parts, dim=dim)
        shape = list(p0.shape)
        shape.insert(dim, len(parts))
        key = (tuple(shape), p0.dtype)
        ring = self.slots.get(key)
        if ring is None:
            # all buffers of a shape are pinned when it is first seen (page-locking costs milliseconds per buffer: done lazily, the first
            # `depth` training steps would each pay for it)
            ring = self.slots[key] = {'i': 0, 'bufs': [torch.empty(shape, dtype=p0.dtype).pin_memory() for _ in range(self.depth)],
                                      'evs': [None] * self.depth}
            if len(self.slots) > 16:                                  # varying batch shapes: forget the oldest shape
                self.slots.pop(next(iter(self.slots)))
        else:
            self.slots[key] = self.slots.pop(key)                     # most recently used last: the shape forgotten above is the stalest
        i = ring['i']
        ring['i'] = (i + 1) % self.depth
        if ring['evs'][i] is not None:
            ring['evs'][i].synchronize()
        buf = ring['bufs'][i]
        # numpy does the interleave on the calling thread: torch's CPU stack / reductions wake the whole intra-op pool, which costs
        # milliseconds per call when the workers have gone to sleep between training steps (128-thread host)
        bn = buf.numpy()
        for j, part in enumerate(parts):
            bn[(slice(None),) * dim + (j,)] = part.numpy()
        # the copy runs on a side stream: on the compute stream it could only start after every kernel already queued there (the previous
        # training step), and the step's first kernel only after it -- 0.1-0.15 ms per step exposed for a 4 MB batch
        cur = torch.cuda.current_stream()
        cs = _copy_stream(cur.device) if _COPY_STREAM else cur
        with torch.cuda.stream(cs):
            out = buf.to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
        cur.wait_event(ev)
        out.record_stream(cur)                # allocated on the side stream's pool, consumed on the compute stream
        ring['evs'][i] = ev
        return out, buf


_pinned = _PinnedRing()
_copy_streams = {}
_COPY_STREAM = os.environ.get('NR_COPY_STREAM', '1') == '1'       # A/B knob: 0 = batch copies on the compute stream


def _copy_stream(device):
    st = _copy_streams.get(device)
    if st is None:
        st = _copy_streams[device] = torch.cuda.Stream(device=device)
    return st


def stack_to_device(parts, device, num_rows=None, what="index"):
    """1+K (or N) per-position id tensors [B, ...] -> one device tensor [B, len(parts), ...]; host-resident ids are range-checked
    (check_ids) on the stacked host buffer."""
    r = _pinned.stage(parts, device)
    if isinstance(r, tuple):
        out, host = r
        if num_rows is not None:
            check_ids(host, num_rows, what)
        return out
    if num_rows is not None:
        check_ids(r, num_rows, what)
    return r


def profiling(prefix):
    """True when an active ``profile`` would record launches whose name starts with ``prefix`` (callers that can issue a whole
    sequence of launches from one C call fall back to the per-launch form only then)."""
    rec = _prof
    if rec is None:
        return False
    return rec.only is None or any(n.startswith(prefix) for n in rec.only)


# launches issued by the most recent whole-sequence calls, for callers that time such a call as one event pair (bench.py)
seq_launches = {}


def _timed(name, fn):
    rec = _prof
    if rec is None or (rec.only is not None and name not in rec.only):
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    rec.events.setdefault(name, []).append((e0, e1))
    return r


def _call(name, fn, *args):
    _timed(name, lambda: _ck(fn(*args)))


def _ptr(t):
    return None if t is None else t.data_ptr()


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def tuned_dims(d_model, heads, qdim=None, length=None):
    """Whether the TUNED kernels take this geometry: word_embedding_dim (or the pooled width) 300, num_attention_heads 15, query_vector_dim
    <= 208, sequence length <= 50 (src/config.py:21-22,34,39,45).  Everything else runs on the general-geometry path (ops_generic.py,
    csrc/k_generic.h): same results, not tuned."""
    return (d_model == NR_D and heads == NR_HEADS and (qdim is None or 0 < qdim <= NR_QP) and (length is None or 1 <= length <= 50))


_CHECK_DEVICE_IDS = os.environ.get('NR_CHECK_IDS', '0') == '1'


def check_ids(ids, num_rows, what="index"):
    """nn.Embedding raises IndexError on an id outside [0, num_rows) (the kernels clamp instead of faulting).  Host-resident id
    tensors -- what the reference's DataLoader delivers (train.py:202) -- are always checked (a cheap CPU min/max before the H2D copy);
    device-resident ones only with NR_CHECK_IDS=1, because the check costs a stream synchronisation per call."""
    if ids.numel() == 0 or (ids.is_cuda and not _CHECK_DEVICE_IDS):
        return
    if ids.is_cuda:
        lo, hi = int(ids.min()), int(ids.max())
    else:
        # numpy on the host: torch's CPU reductions wake the whole intra-op thread pool (measured 11 ms per call on a 128-thread host
        # whose workers had gone to sleep between training steps; numpy scans 4.3 MB in ~0.2 ms on one core)
        a = ids.detach().numpy()
        lo, hi = int(a.min()), int(a.max())
    if lo < 0 or hi >= num_rows:
        raise IndexError(f"{what} out of range: got [{lo}, {hi}], table has {num_rows} rows")


def padded_len(L, what="sequence length"):
    """Instantiated sequence length that holds a sequence of L items: 20 or 50 (the kernels are templated on those).  Shorter
    sequences are zero-padded by the host and run with key lengths / a pooling length of L, so that any num_words_title /
    num_clicked_news_a_user in [1, 50] works (src/config.py:21-22)."""
    if L < 1 or L > 50:
        raise NotImplementedError(f"{what} must be in [1, 50] (got {L}): the HIP kernels are instantiated for 20 and 50 positions")
    return 20 if L <= 20 else 50


_len_cache = {}


def uniform_lengths(n, L, device):
    """int32 [n] filled with L on `device` (cached)."""
    k = (n, L, str(device))
    t = _len_cache.get(k)
    if t is None:
        if len(_len_cache) > 64:
            _len_cache.clear()
        t = _len_cache[k] = torch.full((n,), L, dtype=torch.int32, device=device)
    return t


def new_seed():
    """Seed for the kernels' counter-based dropout RNG, drawn from torch's CPU generator (torch.manual_seed-able)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


# ----------------------------------------------------------------------------------------------------------
# weight packing (fp32 parameters -> zero-padded bf16 MFMA operands) of the LIVE parameter storage the optimizer
# updates in place: re-packed whenever the parameters changed, see _packed()
# ----------------------------------------------------------------------------------------------------------
def untile(t, R, K):
    """Packed weight operands are stored in the kernels' "tile order" (include/nr_engine.h): 16 x 32 blocks of 64 lane fragments.
    Returns the row-major [R, K] matrix (a copy) for the library GEMMs that take the same weights."""
    return t.view(R // 16, K // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(R, K)


# Packed operands are cached per parameter STATE: the source tensor objects, their storage pointers and version counters + a global epoch.
# torch's own in-place updates (optimizer.step(), load_state_dict, .copy_) bump the version counter; code that writes parameter memory
# behind torch's back -- the engine's fused Adam kernel, collectives on ``p.data`` -- calls ``invalidate_packed()``.  A training step packs
# each operand once (not once per encoder call plus once per backward), evaluation packs once per model state.
_pack_cache = collections.OrderedDict()
_pack_epoch = 0
_PACK_CACHE_MAX = 64


def invalidate_packed():
    """Parameters were modified without torch noticing (raw kernel writes / ``.data`` writes): drop every cached packed operand."""
    global _pack_epoch
    _pack_epoch += 1
    _pack_cache.clear()


def _packed(kind, tensors, build):
    """Cached build() keyed on the source tensor OBJECTS (weak references: an address / id recycled by a later model never matches), their
    storage pointers and version counters, and the global epoch."""
    key = (kind,) + tuple(id(t) for t in tensors)
    state = tuple((t.data_ptr(), t._version) for t in tensors)
    hit = _pack_cache.get(key)
    if hit is not None:
        refs, st, epoch, out = hit
        if epoch == _pack_epoch and st == state and all(r() is t for r, t in zip(refs, tensors)):
            _pack_cache.move_to_end(key)
            return out
    out = build()
    _pack_cache[key] = (tuple(weakref.ref(t) for t in tensors), state, _pack_epoch, out)
    _pack_cache.move_to_end(key)
    while len(_pack_cache) > _PACK_CACHE_MAX:
        _pack_cache.popitem(last=False)
    return out


def _packed_fresh(kind, tensors):
    """Whether the cache holds a current entry for (kind, tensors) -- same test as _packed."""
    hit = _pack_cache.get((kind,) + tuple(id(t) for t in tensors))
    if hit is None:
        return False
    refs, st, epoch, _ = hit
    return epoch == _pack_epoch and st == tuple((t.data_ptr(), t._version) for t in tensors) and all(r() is t for r, t in zip(refs, tensors))


def _packed_put(kind, tensors, out):
    _pack_cache[(kind,) + tuple(id(t) for t in tensors)] = (tuple(weakref.ref(t) for t in tensors), tuple((t.data_ptr(), t._version) for t in tensors),
                                                            _pack_epoch, out)
    _pack_cache.move_to_end((kind,) + tuple(id(t) for t in tensors))
    while len(_pack_cache) > _PACK_CACHE_MAX:
        _pack_cache.popitem(last=False)


def prepack_encoder(Wq, bq, Wk, bk, Wv, bv, Wa, ba, qv, kinds):
    """All stale packed operands of ONE encoder in ONE launch (nr_pack_encoder): kinds is a subset of ('qkv', 'qkv32', 'qkv_dx', 'additive',
    'additive_t').  After an optimiser step every operand of the encoder is stale, and packing them one by one costs five ~8 us launches per
    encoder and step; the single pack_* functions below then find their entries in the cache.  Nothing to do when everything asked for is
    current."""
    keys = {'qkv': (Wq, bq, Wk, bk, Wv, bv), 'qkv32': (Wq, bq, Wk, bk, Wv, bv), 'qkv_dx': (Wq, Wk, Wv), 'additive': (Wa, ba, qv), 'additive_t': (Wa,)}
    stale = [k for k in kinds if not _packed_fresh(k, keys[k])]
    if len(stale) < 2:
        return
    dev = Wq.device
    i16, f32 = _BF16_AS_I16, torch.float32
    o = {}
    if 'qkv' in stale:
        o['qkv'] = (torch.empty(3 * NR_NP, NR_KP, dtype=i16, device=dev), torch.empty(3 * NR_NP, dtype=f32, device=dev))
    if 'qkv32' in stale:
        o['qkv32'] = (torch.empty(3 * NR_NP * NR_K16 * 16, dtype=i16, device=dev), torch.empty(3 * NR_NP, dtype=f32, device=dev))
    if 'qkv_dx' in stale:
        o['qkv_dx'] = torch.empty(60 * 10 * 64 * 8, dtype=i16, device=dev)
    if 'additive' in stale:
        o['additive'] = (torch.empty(NR_QP, NR_KP, dtype=i16, device=dev), torch.empty(NR_QP, dtype=f32, device=dev), torch.empty(NR_QP, dtype=f32, device=dev))
    if 'additive_t' in stale:
        o['additive_t'] = torch.empty(NR_KP, 224, dtype=i16, device=dev)
    a = [_f32c(t) for t in (Wq, bq, Wk, bk, Wv, bv, Wa, ba, qv)]
    g = lambda k, i=None: None if k not in o else _ptr(o[k] if i is None else o[k][i])
    _call('nr_pack_encoder', _lib().nr_pack_encoder, *[_ptr(t) for t in a], Wa.shape[0], g('qkv', 0), g('qkv', 1), g('qkv32', 0), g('qkv32', 1),
          g('qkv_dx'), g('additive', 0), g('additive', 1), g('additive', 2), g('additive_t'), _stream())
    for k in stale:
        _packed_put(k, keys[k], o[k])


def pack_qkv(Wq, bq, Wk, bk, Wv, bv):
    def build():
        dev = Wq.device
        Wp = torch.empty(3 * NR_NP, NR_KP, dtype=_BF16_AS_I16, device=dev)
        bp = torch.empty(3 * NR_NP, dtype=torch.float32, device=dev)
        args = [_f32c(t) for t in (Wq, bq, Wk, bk, Wv, bv)]
        _call('nr_pack_qkv', _lib().nr_pack_qkv, *[_ptr(a) for a in args], _ptr(Wp), _ptr(bp), _stream())
        return Wp, bp
    return _packed('qkv', (Wq, bq, Wk, bk, Wv, bv), build)


def pack_qkv32(Wq, bq, Wk, bk, Wv, bv):
    """The projection operand of nr_qkv_proj_fwd: bf16 [3*NP][304] in tile32 order (include/nr_engine.h) + the packed bias vector."""
    def build():
        dev = Wq.device
        Wp32 = torch.empty(3 * NR_NP * NR_K16 * 16, dtype=_BF16_AS_I16, device=dev)
        bp = torch.empty(3 * NR_NP, dtype=torch.float32, device=dev)
        args = [_f32c(t) for t in (Wq, bq, Wk, bk, Wv, bv)]
        _call('nr_pack_qkv32', _lib().nr_pack_qkv32, *[_ptr(a) for a in args], _ptr(Wp32), _ptr(bp), _stream())
        return Wp32, bp
    return _packed('qkv32', (Wq, bq, Wk, bk, Wv, bv), build)


def pack_qkv_dx(Wq, Wk, Wv):
    """[Wq; Wk; Wv] as the fragment-ordered operand of nr_dx_gemm (include/nr_engine.h)."""
    def build():
        WdX = torch.empty(60 * 10 * 64 * 8, dtype=_BF16_AS_I16, device=Wq.device)
        args = [_f32c(t) for t in (Wq, Wk, Wv)]
        _call('nr_pack_qkv_dx', _lib().nr_pack_qkv_dx, *[_ptr(a) for a in args], _ptr(WdX), _stream())
        return WdX
    return _packed('qkv_dx', (Wq, Wk, Wv), build)


def pack_additive(Wa, ba, qv):
    def build():
        dev = Wa.device
        qdim = Wa.shape[0]
        Wap = torch.empty(NR_QP, NR_KP, dtype=_BF16_AS_I16, device=dev)
        bap = torch.empty(NR_QP, dtype=torch.float32, device=dev)
        qvp = torch.empty(NR_QP, dtype=torch.float32, device=dev)
        a = [_f32c(Wa), _f32c(ba), _f32c(qv)]
        _call('nr_pack_additive', _lib().nr_pack_additive, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), qdim, _ptr(Wap), _ptr(bap), _ptr(qvp), _stream())
        return Wap, bap, qvp
    return _packed('additive', (Wa, ba, qv), build)


def pack_additive_t(Wa):
    """Wa^T as the bf16 [KP][QKP] operand of the fused input-gradient product inside nr_additive_bwd_ex."""
    def build():
        WaT = torch.empty(NR_KP, 224, dtype=_BF16_AS_I16, device=Wa.device)
        a = _f32c(Wa)
        _call('nr_pack_additive_t', _lib().nr_pack_additive_t, _ptr(a), Wa.shape[0], _ptr(WaT), _stream())
        return WaT
    return _packed('additive_t', (Wa,), build)


def _bf16(t_i16):
    return t_i16.view(torch.bfloat16)


_zeros16 = {}


# Every weight / input / recurrent product of the backward and recurrent paths runs in the engine's own ring GEMMs (csrc/k_gemm.h, csrc/k_proj.h).
# The hipBLASLt routes they replaced (round 3: chunked batched GEMMs through torch) were kept behind NR_GEMM_HAND=0 for one round of A/B
# (profiles/r04_ab_gemm.txt: conv taps 603 vs 1,254 us, projection gradients 290 vs 358 us, GRU products 165-188 vs 163-230 us) and are gone.


def gemm_nt(A, B, M, N, K, name, out=None, ldc=None):
    """C f32 [M, ldc] = A[:M, :K] @ B[:N, :K]^T for int16-typed bf16 matrices (row strides = their second dims), hand-written (nr_gemm_nt)."""
    ldc = N if ldc is None else ldc
    C = torch.empty(M, ldc, dtype=torch.float32, device=A.device) if out is None else out
    _call(name, _lib().nr_gemm_nt, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(C), ldc, M, N, K, _stream())
    return C


def gemm_tn_parts(G, M, X, ncol, name, taps=1, n_tok=None):
    """G[:, :M]^T @ X[:, :ncol] (taps = 3: the three shifted products of a seqpad X side by side) as fp32 partials [P, M, taps * ncol] over P token
    partitions (nr_gemm_tn); the sum over dim 0 is the gradient."""
    lib = _lib()
    n = G.shape[0] if n_tok is None else n_tok
    N = taps * ncol
    P = lib.nr_gemm_tn_parts(M, N, n)
    out = torch.empty(P, M, N, dtype=torch.float32, device=G.device)
    z = _zeros16.get(G.device)
    if z is None:
        z = _zeros16[G.device] = torch.zeros(64, dtype=_BF16_AS_I16, device=G.device)
    _call(name, lib.nr_gemm_tn, _ptr(G), G.stride(0), M, _ptr(X), X.stride(0), ncol, taps, _ptr(z), _ptr(out), N, n, P, _stream())
    return out


def sum_parts(parts, name='nr_sum_parts'):
    """Sum of the split-K partials over dim 0 in a fixed order (deterministic), one launch."""
    P = parts.shape[0]
    n = parts[0].numel()
    if n % 4:
        return parts.sum(dim=0)
    out = torch.empty(parts.shape[1:], dtype=torch.float32, device=parts.device)
    _call(name, _lib().nr_sum_parts, _ptr(parts), P, n, _ptr(out), 0, _stream())
    return out


# NR_POOL_FLAT: 1 (default) = the pooling backward over the flat token stream (csrc/k_pool3.h: persistent kernel, Wa resident in LDS, any
# sequence length); 0 = the sequence-shaped kernels of rounds 1-3 (A/B only)
_POOL_FLAT = os.environ.get('NR_POOL_FLAT', '1') == '1'
_POOL_FLAT_MIN_TOK = int(os.environ.get('NR_POOL_FLAT_MIN_TOK', '98304'))


NR_POOL_FLAT_QMAX = 200      # rows of Wa the flat kernel keeps in LDS (Pool3Geom::WROWS, csrc/k_pool3.h)


# NR_POOL_FLAT_VIEWS: 1 (default) = sequences of 4 .. 6 positions (NAML's view level) take the flat pooling backward too (round 6); 0 = the
# sequence-shaped kernel (A/B)
_POOL_FLAT_SHORT = os.environ.get('NR_POOL_FLAT_VIEWS', '1') == '1'


def pool_flat_ok(S, act, n_seq=None, *, qdim):
    """Whether the flat kernel takes a pooling level of S-token sequences (act: with the fused activation gradient): 48 consecutive tokens must
    belong to at most 16 sequences (4 with act) -- since round 6 that includes the final attention over NAML's 4 views -- and the
    batch must be worth a persistent launch (one workgroup per CU loads the projection matrix once: 512 click histories are faster on the
    sequence-shaped kernel, 36 vs 47 us).  qdim (query_vector_dim, keyword-only so that no call site can forget it): the flat kernel holds
    200 rows of the projection matrix in LDS; 201 .. 208 stay on the sequence-shaped kernels, which keep all 208 packed rows."""
    return (_POOL_FLAT and qdim <= NR_POOL_FLAT_QMAX and S >= (16 if act else (4 if _POOL_FLAT_SHORT else 7))
            and (n_seq is None or n_seq * S >= _POOL_FLAT_MIN_TOK))


# NR_POOL_FWD_FLAT: 1 (default) = the pooling FORWARD of large launches over whole sequences per wave (csrc/k_pool4.h); 0 = the LDS-tile kernel (A/B)
_POOL_FWD_FLAT = os.environ.get('NR_POOL_FWD_FLAT', '1') == '1'


def pool_fwd_flat_ok(S, n_seq, *, qdim):
    """Whether the persistent whole-sequence forward takes a pooling level: 16 <= S <= 64, query_vector_dim <= 200 (the rows it keeps in LDS), and
    enough tokens to be worth one workgroup per CU loading the projection matrix (the same threshold as the flat backward)."""
    return _POOL_FWD_FLAT and 16 <= S <= 64 and qdim <= NR_POOL_FLAT_QMAX and n_seq * S >= _POOL_FLAT_MIN_TOK


def pool_bwd_flat(ctx_b, Wap, bap, qvp, aw, g, y_ptr, y_stride, n_seq, S, qdim, tag, want_dctx=True, dy=None, p_drop=0.0, ws_tag='', g_stride=NR_D,
                  dq_ring=False):
    """nr_additive_bwd_flat on workspaces: returns (dpre bf16 [n_seq*S][QP], dq_part f32 [grid][QP], dgemm bf16 [n_seq*S][KP] or None).
    y_ptr / y_stride: the pooled vectors of the forward (f32 rows).  dy: seqpad gradient buffer of a conv text encoder -> the fused
    activation gradient goes there instead of dgemm.  g_stride: row stride of g in floats (a column block of wider rows is read in place)."""
    lib = _lib()
    dev = ctx_b.device
    ntok = n_seq * S
    nwg = lib.nr_additive_bwd_flat_grid(ntok)
    dpre = _workspace(f'dpre[{ws_tag}]' if ws_tag else 'dpre', (ntok, NR_QP), _BF16_AS_I16, dev)
    dq_part = dq_slot(nwg, dev) if dq_ring else None          # (its sum may wait for the end of the backward pass: PartSum)
    if dq_part is None:
        dq_part = _workspace(f'dqp[{ws_tag}]' if ws_tag else 'dqp', (nwg, NR_QP), torch.float32, dev)
    tot = _workspace('pool_tot', (n_seq,), torch.float32, dev)
    dgemm = _workspace(f'dctx[{tag}]', (ntok, NR_KP), _BF16_AS_I16, dev) if (want_dctx and dy is None) else None
    _call(f'nr_additive_bwd[{tag}]', lib.nr_additive_bwd_flat_gs, _ptr(ctx_b), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(aw), _ptr(g), g_stride, y_ptr,
          y_stride, _ptr(tot), _ptr(dpre), _ptr(dq_part), _ptr(dgemm), _ptr(dy), p_drop, n_seq, S, qdim, _stream())
    return dpre, dq_part, dgemm


_WGRAD_UNPACK = os.environ.get('NR_WGRAD_UNPACK', '1') == '1'       # A/B knob: 0 = hand the nine weight gradients to autograd

# Weight-gradient phases postponed past the end of loss.backward() (see _EncoderFn.backward): set by a trainer whose gradient buffers are
# persistent (optim.EngineAdam under a process group); the trainer calls run_deferred() before it reads any weight gradient.
defer_wgrad = False
_deferred = []


def run_deferred():
    """Run (and forget) the postponed weight-gradient phases of the backward passes since the last call, in their backward order; then hand
    the gradients they (or a backward outside autograd's engine) queued to their buffers (flush_grads)."""
    while _deferred:
        _deferred.pop(0)()
    flush_grads()


def drop_deferred():
    global _dq_outstanding
    _deferred.clear()
    _gq.clear()
    _dq_outstanding = 0


# ---- small weight gradients of a backward pass -> the trainer's persistent gradient buffers, in ONE launch ---------------------------------
# autograd hands every gradient a backward returns to an AccumulateGrad node: one element-wise add per parameter (21 launches per NAML step,
# 10 per LSTUR step, each a few microseconds of work behind a launch).  When the trainer owns a persistent buffer for every parameter of a group
# (inplace_grads), the backward functions queue (buffer, gradient view) pairs here instead and return None; the queue is flushed by one
# nr_accum_many launch when autograd's engine finishes the pass (Engine.queue_callback), i.e. before loss.backward() returns.
_gq = []
_gq_armed = False


class _AccumItem(_capi.ctypes.Structure):
    _fields_ = [('src', _capi.ctypes.c_void_p), ('dst', _capi.ctypes.c_void_p), ('src_ld', _capi.ctypes.c_int64), ('dst_ld', _capi.ctypes.c_int64),
                ('rows', _capi.ctypes.c_int32), ('cols', _capi.ctypes.c_int32), ('parts', _capi.ctypes.c_int32), ('reserved', _capi.ctypes.c_int32),
                ('part_stride', _capi.ctypes.c_int64)]


class PartSum:
    """parts[:, :cols].sum(0) of a persistent kernel's per-workgroup partial rows (f32 [P, ld]: the query-vector gradient of a pooling level),
    not evaluated yet: queue_grad() folds the sum into the backward pass's one accumulate launch; materialize() runs it now (nr_sum_parts)."""

    def __init__(self, parts, cols):
        self.parts, self.cols = parts, int(cols)
        self.shape = (self.cols,)

    def materialize(self):
        return sum_parts(self.parts)[:self.cols]


# per-workgroup partial rows that wait for the end of the backward pass must not be overwritten by the next pooling level's: a ring of small
# scratch buffers, one per PartSum outstanding (a backward pass holds at most a handful: title, abstract, views, user)
_DQ_SLOTS = 12
_dq_next = 0
_dq_outstanding = 0


def dq_slot(nwg, device):
    """f32 [nwg, NR_QP] scratch for a pooling level's query-vector partials, or None when the ring is exhausted (the caller then uses its shared
    workspace and sums at once)."""
    global _dq_next, _dq_outstanding
    if _dq_outstanding >= _DQ_SLOTS or not _PART_SUM:
        return None
    _dq_outstanding += 1
    k = _dq_next
    _dq_next = (k + 1) % _DQ_SLOTS
    t = _workspace(f'dqp#{k}', (nwg, NR_QP), torch.float32, device)
    t._nr_dq_ring = True
    return t


def is_dq_slot(t):
    return getattr(t, '_nr_dq_ring', False)


_PART_SUM = os.environ.get('NR_PART_SUM', '1') == '1'      # A/B knob: 0 = every query-vector gradient through its own nr_sum_parts launch


def _accum_geometry(t):
    """(rows, cols, ld) of a 1-D / 2-D float32 view whose rows are contiguous, or None."""
    if t.dim() == 1:
        n = t.shape[0]
        return (1, n, n) if (n <= 1 or t.stride(0) == 1) else (n, 1, t.stride(0))
    if t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1):
        return (t.shape[0], t.shape[1], max(t.stride(0), t.shape[1]) if t.shape[0] > 1 else t.shape[1])
    return None


def queue_grad(dst, src):
    """dst += src at the end of the running backward pass (or at the next flush_grads()).  dst: a view of a persistent gradient buffer, src: the
    gradient (any float32 view of the same shape; 1-D / 2-D views with contiguous rows go in as they are, anything else through a copy)."""
    if isinstance(src, PartSum):
        if dst.dim() != 1 or dst.shape[0] != src.cols or not dst.is_contiguous() or src.parts.stride(1) != 1:
            src = src.materialize()
        else:
            _gq.append((dst, src))
            _arm_flush()
            return
    if src.dtype != torch.float32:
        src = src.to(torch.float32)
    if tuple(dst.shape) != tuple(src.shape):
        src = src.reshape(dst.shape)
    if dst.dim() > 2 or _accum_geometry(dst) is None:
        if not dst.is_contiguous():
            raise ValueError("queue_grad: the destination must be a contiguous buffer or a 1-D / 2-D view with contiguous rows")
        dst = dst.view(-1, dst.shape[-1]) if dst.dim() > 1 else dst
        src = src.reshape(dst.shape)
    if _accum_geometry(src) is None:
        src = src.contiguous()
    _gq.append((dst, src))
    _arm_flush()


def _arm_flush():
    global _gq_armed
    if _gq_armed:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_flush_from_engine)
        _gq_armed = True
    except RuntimeError:
        pass                 # not inside a backward pass (a postponed phase run by run_deferred(), a direct call): the caller flushes


def _flush_from_engine():
    global _gq_armed
    _gq_armed = False
    flush_grads()


def flush_grads():
    """Issue the queued accumulations now (one launch per 48 items; items that share a destination go into successive launches)."""
    global _gq_armed, _dq_outstanding
    _dq_outstanding = 0                        # (the ring's rows are read by the launches below, in stream order before anything can refill them)
    if not _gq:
        return
    todo = list(_gq)
    _gq.clear()
    while todo:
        seen, now, rest = set(), [], []
        for d, s_ in todo:
            key = d.data_ptr()
            (rest if key in seen else now).append((d, s_))
            seen.add(key)
        arr = (_AccumItem * len(now))()
        for i, (d, s_) in enumerate(now):
            if isinstance(s_, PartSum):
                arr[i] = _AccumItem(s_.parts.data_ptr(), d.data_ptr(), s_.cols, s_.cols, 1, s_.cols, s_.parts.shape[0], 0, s_.parts.stride(0))
                continue
            rd, cd, ldd = _accum_geometry(d)
            rs, cs, lds = _accum_geometry(s_)
            if (rd, cd) != (rs, cs):                 # same elements, different 2-D reading (a contiguous vector against a column): one of them is [n] x 1
                if rd * cd != rs * cs or 1 not in (cd, cs):
                    raise ValueError("queue_grad: incompatible views")
                if cd != 1:
                    rd, cd, ldd = cd, 1, 1
                if cs != 1:
                    rs, cs, lds = cs, 1, 1
            arr[i] = _AccumItem(s_.data_ptr(), d.data_ptr(), lds, ldd, rd, cd, 1, 0, 0)
        _call('nr_accum_many', _lib().nr_accum_many, _capi.ctypes.cast(arr, _capi.ctypes.c_void_p), len(now), _stream())
        todo = rest


def hand_over_grads(params, vals):
    """The values a backward returns for ``params``: the gradients themselves for plain autograd; (None, ...) after queueing them for the
    trainer's persistent buffers when it owns one for every parameter (inplace_grads)."""
    dst = inplace_grads(params)
    if dst is None:
        return tuple(v.materialize() if isinstance(v, PartSum) else v for v in vals)
    for d, v in zip(dst, vals):
        queue_grad(d, v)
    return (None,) * len(params)


def inplace_grads(params):
    """The persistent gradient buffers of ``params`` when a trainer owns one for EVERY one of them (``_nr_inplace_grad``, see
    grad_target), else None.  The encoder backward then accumulates its weight gradients there itself (nr_wgrad_unpack: one launch)
    and hands autograd None for them, instead of 9 strided slices that AccumulateGrad adds one by one."""
    if not _WGRAD_UNPACK:
        return None
    out = []
    for p in params:
        g = getattr(p, 'grad', None)
        if not (getattr(p, '_nr_inplace_grad', False) and g is not None and g.dtype == torch.float32 and g.is_contiguous()
                and g.shape == p.shape and g.device == p.device):
            return None
        out.append(g)
    return out


# NR_FWD_SPLIT: 1 (default) = title encoders that need gradients run nr_qkv_proj_fwd + nr_attn_fwd (csrc/k_proj.h) instead of the
# register-resident nr_mhsa_fwd kernel; 2 = inference too; 0 = never (A/B)
_FWD_SPLIT = int(os.environ.get('NR_FWD_SPLIT', '1'))

_ws = {}
_side = {}
def grad_target(param):
    """Where a large table gradient goes.  Default: a fresh zeroed tensor handed back to autograd (which then adds it into
    ``param.grad``: for the 85 MB word table that is a 85 MB fill plus a 255 MB read-add-write per step, 180 / 540 MB for LSTUR's user
    table).  A trainer that owns persistent gradient buffers (``dist.FlatGradBuffer``: explicit zero() and all-reduce, no autograd
    hooks on the parameter) marks its parameters with ``_nr_inplace_grad``; the scatter kernels (which accumulate) then write straight
    into ``param.grad`` and the backward returns None for the table.  Returns (tensor to accumulate into, value to return)."""
    g = getattr(param, 'grad', None)
    if (getattr(param, '_nr_inplace_grad', False) and g is not None and g.dtype == torch.float32 and g.is_contiguous()
            and g.shape == param.shape and g.device == param.device):
        return g, None
    d = torch.zeros(param.shape, dtype=torch.float32, device=param.device)
    return d, d


def table_grad_ready(param):
    """Tell the owner of the gradient buffers (optim.EngineAdam) that this table's gradient is complete on the current stream."""
    cb = getattr(param, '_nr_grad_ready', None)
    if cb is not None:
        cb()


def sort_ids_async(ids, num_rows=None):
    """Sort the token ids for the embedding backward on a side HIP stream, overlapped with the forward kernels (the ids are
    known before the forward starts; the sorted order is only needed by the scatter at the very end of the backward).
    ``nr_sort_ids``: stable LSD radix sort on the ceil(log2(num_rows)) significant bits (2 passes for the 70,976- and 130,001-word
    vocabularies) instead of torch.sort's 64-bit merge sort.  Returns (ids_sorted, perm, event)."""
    dev = ids.device
    cur = torch.cuda.current_stream(dev)
    st = _side.get(dev)
    if st is None:
        st = _side[dev] = torch.cuda.Stream(device=dev)
    st.wait_stream(cur)
    with torch.cuda.stream(st):
        flat = ids.reshape(-1)
        n = flat.numel()
        lib = _lib()
        ws_bytes = lib.nr_sort_ids_workspace(n, num_rows) if num_rows is not None else -1
        if ws_bytes < 0:
            raise NotImplementedError(f"nr_sort_ids: {n} ids over {num_rows} rows is outside the supported range")
        ids_sorted = torch.empty(n, dtype=torch.int64, device=dev)
        perm = torch.empty(n, dtype=torch.int64, device=dev)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
        _call('nr_sort_ids', lib.nr_sort_ids, _ptr(flat), n, num_rows, _ptr(ids_sorted), _ptr(perm), _ptr(ws), ws_bytes, st.cuda_stream)
        ev = torch.cuda.Event()
        ev.record(st)
    ids.record_stream(st)
    return ids_sorted, perm, ev


def sort_ids(ids, num_rows):
    """nr_sort_ids on the current stream: (ids_sorted, perm) of a flat int64 id tensor, stable."""
    flat = ids.reshape(-1).contiguous()
    n = flat.numel()
    lib = _lib()
    ws_bytes = lib.nr_sort_ids_workspace(n, num_rows)
    if ws_bytes < 0:
        raise NotImplementedError(f"nr_sort_ids: {n} ids over {num_rows} rows is outside the supported range")
    ids_sorted = torch.empty(n, dtype=torch.int64, device=flat.device)
    perm = torch.empty(n, dtype=torch.int64, device=flat.device)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=flat.device)
    _call('nr_sort_ids', lib.nr_sort_ids, _ptr(flat), n, num_rows, _ptr(ids_sorted), _ptr(perm), _ptr(ws), ws_bytes, _stream())
    return ids_sorted, perm


def sorted_ids_ready(pack):
    """Make the current stream wait for sort_ids_async's result and hand the tensors over to it."""
    ids_sorted, perm, ev = pack
    cur = torch.cuda.current_stream(ids_sorted.device)
    cur.wait_event(ev)
    ids_sorted.record_stream(cur)
    perm.record_stream(cur)
    return ids_sorted, perm


_WS_ZERO_SHAPES = 4


# ---- buffers that live from a forward to its backward and rely on an initial fill of the regions no kernel writes -------------------------------
# (the GRU's state history: padding columns; a text encoder's seqpad token store: the row before the first sequence and the tail rows).  The buffer
# of the previous step is handed out again -- no fill -- once its backward has released it; a forward that arrives while it is still owed to a
# backward (two forwards in flight, a forward whose backward never runs) gets a fresh, initialised one, which then becomes the pooled one.
_step_pool = {}


def step_buffer(key, shape, dtype, device, init):
    """init(t): fills a fresh tensor's never-written regions.  Returns a tensor of `shape` whose kernel-written regions hold stale data."""
    k = (key, tuple(int(x) for x in shape), dtype, str(device))
    e = _step_pool.get(k)
    if e is not None and not e[1]:
        e[1] = True
        return e[0]
    t = torch.empty(k[1], dtype=dtype, device=device)
    init(t)
    if len(_step_pool) > 16:
        _step_pool.clear()
    _step_pool[k] = [t, True]
    return t


def step_buffer_release(t):
    """The backward's last reader of `t` has been enqueued: the next forward on this stream may overwrite it."""
    if t is None:
        return
    for e in _step_pool.values():
        if e[0] is t or (e[0].data_ptr() == t.data_ptr() and e[0].numel() == t.numel()):
            e[1] = False


def _workspace(key, shape, dtype, device, zero=False):
    """Reusable scratch buffers that live only inside one backward call (all users are ordered on the current stream).
    Plain buffers: ONE flat allocation per (key, dtype, device), grown to the largest request and sliced -- a cache keyed on the shape
    would keep a buffer per distinct batch shape forever.  ``zero=True`` buffers rely on the columns / rows the kernels never write
    staying zero, so they are kept per shape (zeroed once when created), at most _WS_ZERO_SHAPES shapes per key, least recently used
    evicted: a training step alternates between two shapes (user encoder, news encoder) and must not pay a fill per call."""
    shape = tuple(int(x) for x in shape)
    k = (key, dtype, str(device))
    if zero:
        lru = _ws.setdefault(k, collections.OrderedDict())
        t = lru.get(shape)
        if t is None:
            t = lru[shape] = torch.zeros(shape, dtype=dtype, device=device)
            while len(lru) > _WS_ZERO_SHAPES:
                lru.popitem(last=False)
        else:
            lru.move_to_end(shape)
        return t
    n = 1
    for x in shape:
        n *= x
    buf = _ws.get(k)
    if buf is None or buf.numel() < n:
        buf = _ws[k] = torch.empty(max(n, 1), dtype=dtype, device=device)
    return buf[:n].view(shape)


# ----------------------------------------------------------------------------------------------------------
# fused encoder: [gather | dense] -> MHSA -> additive pooling
# ----------------------------------------------------------------------------------------------------------
class _EncoderFn(torch.autograd.Function):
    """out[n_seq, D] = AdditiveAttention(dropout2(MHSA(dropout1(x)))),  x = table[ids] or a dense [n_seq,S,D] tensor.

    Mirrors src/model/NRMS/news_encoder.py:27-48 (ids form, dropout) and src/model/NRMS/user_encoder.py:15-26
    (dense form, no dropout)."""

    @staticmethod
    def forward(ctx, ids, table, x_dense, Wq, bq, Wk, bk, Wv, bv, Wa, ba, qv, S, p_drop, seed, valid=None):
        """S: instantiated length the inputs are padded to; valid (<= S): real length -- positions >= valid are masked as attention
        keys (nr_mhsa_fwd_len) and excluded from the pooling (nr_additive_fwd_v), so they receive and produce zero gradients."""
        lib = _lib()
        gather = ids is not None
        dev = Wq.device
        n_seq = ids.shape[0] if gather else x_dense.shape[0]
        if not lib.nr_supported_seq_len(S):
            raise NotImplementedError(f"sequence length {S} is not instantiated in the HIP kernels (20, 50)")
        valid = S if valid is None else int(valid)
        key_len = uniform_lengths(n_seq, valid, dev) if valid < S else None
        need_grad = any(ctx.needs_input_grad)
        # training form of the title encoder: gather-fused projection GEMM + stand-alone attention kernel on head-major saves (csrc/k_proj.h);
        # the register-resident kernel stays the inference form (_FWD_SPLIT = 2 uses the split form there too)
        split = gather and S == 20 and (_FWD_SPLIT == 2 or (_FWD_SPLIT == 1 and need_grad))
        flat_bwd = pool_flat_ok(S, False, n_seq, qdim=Wa.shape[0])
        prepack_encoder(Wq, bq, Wk, bk, Wv, bv, Wa, ba, qv, (['qkv32'] if split else ['qkv']) + ['additive'] + (['qkv_dx'] if need_grad else [])
                        + (['additive_t'] if (need_grad and not flat_bwd) else []))
        Wp, bp = (None, None) if split else pack_qkv(Wq, bq, Wk, bk, Wv, bv)
        Wap, bap, qvp = pack_additive(Wa, ba, qv)
        cbuf = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev)
        sp4 = (S + 3) // 4 * 4
        WaT = pack_additive_t(Wa) if (need_grad and not pool_flat_ok(S, False, n_seq, qdim=Wa.shape[0])) else None
        WpT = None
        pooled = False
        if need_grad:
            WpT = pack_qkv_dx(Wq, Wk, Wv)
        if split:
            qs = torch.empty(n_seq * NR_QKV_HM_SEQ, dtype=_BF16_AS_I16, device=dev)      # head-major Q | K | V^T
            ks = vts = None
            xb = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev) if need_grad else None
        elif need_grad:
            qs = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev)
            ks = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev)
            vts = torch.empty(n_seq, NR_HEADS, NR_DK, sp4, dtype=_BF16_AS_I16, device=dev)
            xb = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev)       # masked bf16 tokens for dW = dqkv^T @ X (written by the kernel)
        else:
            qs = ks = vts = xb = None
        if split:
            ids_c = ids.contiguous()
            tab = table.detach()
            assert tab.dtype == torch.float32 and tab.is_contiguous() and tab.shape[1] == NR_D
            Wp32, bp32 = pack_qkv32(Wq, bq, Wk, bk, Wv, bv)
            _call(f'nr_qkv_proj_fwd[S={S}]', lib.nr_qkv_proj_fwd, _ptr(ids_c), _ptr(tab), tab.shape[0], _ptr(Wp32), _ptr(bp32), _ptr(qs), _ptr(xb),
                  n_seq, S, p_drop, seed, _stream())
            # the titles are pooled inside the attention kernel (the ctx tile is pooled from LDS instead of being read back by nr_additive_fwd;
            # bit for bit the two-launch form nr_attn_fwd + nr_additive_fwd, which tests/kernel_checks_proj.py holds it against)
            out = torch.empty(n_seq, NR_D, dtype=torch.float32, device=dev)
            aw = torch.empty(n_seq, S, dtype=torch.float32, device=dev)
            _call(f'nr_attn_pool_fwd[S={S}]', lib.nr_attn_pool_fwd, _ptr(qs), _ptr(cbuf), _ptr(key_len), _ptr(Wap), _ptr(bap), _ptr(qvp),
                  _ptr(out), NR_D, _ptr(aw), n_seq, S, valid, p_drop, seed, _stream())
            pooled = True
            xd = None
        elif gather:
            ids_c = ids.contiguous()
            tab = table.detach()
            assert tab.dtype == torch.float32 and tab.is_contiguous() and tab.shape[1] == NR_D
            _call(f'nr_mhsa_fwd[S={S}]', lib.nr_mhsa_fwd_len, _ptr(ids_c), _ptr(tab), tab.shape[0], None, _ptr(Wp), _ptr(bp), _ptr(cbuf),
                                _ptr(qs), _ptr(ks), _ptr(vts), _ptr(xb), _ptr(key_len), n_seq, S, p_drop, seed, _stream())
            xd = None
        else:
            ids_c = None
            xd = _f32c(x_dense)
            _call(f'nr_mhsa_fwd[S={S}]', lib.nr_mhsa_fwd_len, None, None, 0, _ptr(xd), _ptr(Wp), _ptr(bp), _ptr(cbuf),
                                _ptr(qs), _ptr(ks), _ptr(vts), _ptr(xb), _ptr(key_len), n_seq, S, p_drop, seed, _stream())
        if not pooled:
            out = torch.empty(n_seq, NR_D, dtype=torch.float32, device=dev)
            aw = torch.empty(n_seq, S, dtype=torch.float32, device=dev)
            _call(f'nr_additive_fwd[S={S}]', lib.nr_additive_fwd_v, _ptr(cbuf), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(out), NR_D, None, 0, _ptr(aw),
                  n_seq, S, valid, _stream())
        if need_grad:
            ctx.save_for_backward(ids_c, table if gather else None, xd, cbuf, qs, ks, vts, aw, WpT, Wap, bap, qvp, xb, WaT, key_len, out)
            ctx.meta = (S, p_drop, seed, n_seq, Wa.shape[0], gather, split)
            ctx.table_param = table                 # the caller's tensor object (the nn.Parameter): see grad_target()
            ctx.wparams = (Wq, bq, Wk, bk, Wv, bv, Wa, ba, qv)        # likewise: inplace_grads()
            ctx.sorted = sort_ids_async(ids_c, table.shape[0]) if gather and ctx.needs_input_grad[1] else None
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib()
        ids, table, xd, cbuf, qs, ks, vts, aw, WpT, Wap, bap, qvp, Xb, WaT, key_len, y = ctx.saved_tensors
        S, p_drop, seed, n_seq, qdim, gather, split = ctx.meta
        dev = cbuf.device
        ntok = n_seq * S
        g_out = g_out.to(torch.float32).contiguous()
        # Two phases.  Phase 1 is everything the INPUT gradient needs -- pooling backward, attention backward, dX, the embedding scatter -- and ends
        # with table_grad_ready: under data parallelism the table bucket's all-reduce starts there.  Phase 2, the weight-gradient GEMMs (dWa, dWqkv)
        # and their hand-off, needs nothing from the exchange and is what runs WHILE the table is on the wire: immediately after phase 1, or --
        # when a trainer with persistent gradient buffers asks for it (ops.defer_wgrad: EngineAdam under a process group, graph.SegmentedStep)
        # -- postponed to ops.run_deferred(), which the trainer calls after loss.backward() has returned (a HIP-graph boundary can then sit
        # between the two phases).  Scratch that phase 2 reads (dpre, dqkv) is keyed per encoder shape so that a later encoder's phase 1 cannot
        # overwrite it.
        # ---- phase 1: additive attention backward -> dpre (kernel) --------------------------------------------------------------------------
        if pool_flat_ok(S, False, n_seq, qdim=qdim):
            dpre, dq_part, dctx_gemm = pool_bwd_flat(cbuf, Wap, bap, qvp, aw, g_out, _ptr(y), y.stride(0), n_seq, S, qdim, f'S={S}',
                                                     ws_tag=f'{S},{int(gather)}')
            nwg = dq_part.shape[0]
        else:
            nwg = lib.nr_additive_bwd_grid(n_seq, S)
            dpre = _workspace(f'dpre[{S},{int(gather)}]', (ntok, NR_QP), _BF16_AS_I16, dev)
            dq_part = _workspace(f'dqp[{S},{int(gather)}]', (nwg, NR_QP), torch.float32, dev)
            dctx_gemm = _workspace('dctx', (ntok, NR_KP), _BF16_AS_I16, dev)       # = dpre @ Wa, produced inside the kernel
            _call(f'nr_additive_bwd[S={S}]', lib.nr_additive_bwd_ex, _ptr(cbuf), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(aw), _ptr(g_out), _ptr(dpre),
                  _ptr(dq_part), _ptr(WaT), _ptr(dctx_gemm), n_seq, S, _stream())
        # ---- attention backward (kernel) -> dqkv ------------------------------------------------------------------
        dqkv = _workspace('dqkv', (ntok, NR_LDG), _BF16_AS_I16, dev, zero=True)   # padding columns stay zero; one buffer per shape (news / user encoder)
        if split:
            _call(f'nr_attn_bwd[S={S}]', lib.nr_attn_bwd_hm, _ptr(qs), _ptr(dctx_gemm), NR_KP, _ptr(aw), _ptr(g_out), _ptr(dqkv), _ptr(key_len),
                  n_seq, S, p_drop, seed, _stream())
        else:
            _call(f'nr_attn_bwd[S={S}]', lib.nr_attn_bwd_len, _ptr(qs), _ptr(ks), _ptr(vts), _ptr(dctx_gemm), NR_KP, _ptr(aw), _ptr(g_out), _ptr(dqkv),
                  _ptr(key_len), n_seq, S, p_drop, seed, _stream())
        # ---- input gradient: dX = dqkv @ [Wq; Wk; Wv] (nr_dx_gemm), then the embedding scatter.  The table gradient is the large message of the
        # data-parallel exchange: its all-reduce is started by table_grad_ready on RCCL's stream as soon as the scatter is enqueued -----
        dX = torch.empty(ntok, NR_KP, dtype=torch.bfloat16, device=dev)
        _call(f'nr_dx_gemm[S={S}]', lib.nr_dx_gemm, _ptr(dqkv), _ptr(WpT), _ptr(dX), ntok, _stream())
        d_table = d_x = None
        if gather:
            if ctx.needs_input_grad[1]:
                tdst, d_table = grad_target(ctx.table_param)
                dXi = dX.view(_BF16_AS_I16)
                # sort token ids so that every table row is reduced by adjacent lanes instead of contended atomics
                ids_sorted, perm = sorted_ids_ready(ctx.sorted)
                _call(f'nr_embed_scatter_sorted[S={S}]', lib.nr_embed_scatter_sorted, _ptr(ids_sorted), _ptr(perm), _ptr(dXi), NR_KP,
                      _ptr(tdst), table.shape[0], ntok, p_drop, seed, _stream())
                table_grad_ready(ctx.table_param)
        elif ctx.needs_input_grad[2]:
            # f32 in the layout of x; when x is the history part of split_rows(), straight into its half of the shared gradient buffer
            d_x = grad_dst(xd, (n_seq, S, NR_D)) if xd is not None else None
            if d_x is None:
                d_x = torch.empty(n_seq, S, NR_D, dtype=torch.float32, device=dev)
            _call('nr_rows_to_f32[dx]', lib.nr_rows_to_f32, _ptr(dX), NR_KP, NR_D, _ptr(d_x), NR_D, ntok, _stream())
        # ---- phase 2: the nine weight gradients.  dWa_ext = dpre^T @ [ctx | 1] and dW_ext = dqkv^T @ [X | 1]: split-K ring kernel (csrc/k_gemm.h),
        # partials [P, rows, KP]; column D = bias gradient (ctx[:, D] == X[:, D] == 1).  A trainer with persistent gradient buffers: summed over the
        # partitions and accumulated there in one launch; plain autograd: sums, then slices of the packed geometry that AccumulateGrad adds ------
        dst = inplace_grads(ctx.wparams) if all(ctx.needs_input_grad[3:12]) else None

        def weight_grads():
            dWa_parts = gemm_tn_parts(dpre, NR_QP, cbuf, NR_KP, f'nr_gemm_tn_dWa[S={S}]')
            dW_parts = gemm_tn_parts(dqkv, NR_LDG, Xb, NR_KP, f'nr_gemm_tn_dWqkv[S={S}]')
            return dWa_parts, dW_parts
        if dst is not None:
            def phase2():
                dWa_parts, dW_parts = weight_grads()
                _call('nr_wgrad_unpack', lib.nr_wgrad_unpack, _ptr(dW_parts), dW_parts.shape[0], _ptr(dWa_parts), dWa_parts.shape[0],
                      _ptr(dq_part), nwg, qdim, *[_ptr(g) for g in dst], _stream())
            if defer_wgrad:
                _deferred.append(phase2)
            else:
                phase2()
            return (None, d_table, d_x) + (None,) * 13
        dWa_parts, dW_parts = weight_grads()
        d_qv = sum_parts(dq_part)[:qdim]
        dWa_ext = dWa_parts[0] if dWa_parts.shape[0] == 1 else dWa_parts.sum(dim=0)
        dW_ext = dW_parts[0] if dW_parts.shape[0] == 1 else dW_parts.sum(dim=0)
        d_Wa, d_ba = dWa_ext[:qdim, :NR_D], dWa_ext[:qdim, NR_D]
        gW = [dW_ext[i * NR_KP:i * NR_KP + NR_D, :NR_D] for i in range(3)]
        gb = [dW_ext[i * NR_KP:i * NR_KP + NR_D, NR_D] for i in range(3)]
        return (None, d_table, d_x, gW[0], gb[0], gW[1], gb[1], gW[2], gb[2], d_Wa, d_ba, d_qv, None, None, None, None)


def encode_titles(ids, table, mhsa, additive, p_drop, training):
    """NRMS news encoder on a [n_titles, L] int64 id tensor (already on the GPU)."""
    _require_cuda(table, "word_embedding.weight")
    if not tuned_dims(table.shape[1], mhsa.num_attention_heads, additive.linear.weight.shape[0], ids.shape[1]):
        from . import ops_generic
        return ops_generic.encode_titles(ids, table, mhsa, additive, p_drop, training)
    p = float(p_drop) if training else 0.0
    seed = new_seed() if p > 0 else 0
    L = ids.shape[1]
    S = padded_len(L, "num_words_title")
    if S != L:
        ids = torch.nn.functional.pad(ids, (0, S - L))        # padded positions: masked keys, outside the pooling
    return _EncoderFn.apply(ids, table, None, mhsa.W_Q.weight, mhsa.W_Q.bias, mhsa.W_K.weight, mhsa.W_K.bias,
                            mhsa.W_V.weight, mhsa.W_V.bias, additive.linear.weight, additive.linear.bias,
                            additive.attention_query_vector, S, p, seed, L)


def encode_dense(x, mhsa, additive):
    """NRMS user encoder on a dense [n_seq, S, D] float tensor."""
    _require_cuda(x, "clicked_news_vector")
    if not tuned_dims(x.shape[2], mhsa.num_attention_heads, additive.linear.weight.shape[0], x.shape[1]):
        from . import ops_generic
        return ops_generic.encode_dense(x, mhsa, additive)
    N = x.shape[1]
    S = padded_len(N, "num_clicked_news_a_user")
    if S != N:
        x = torch.nn.functional.pad(x, (0, 0, 0, S - N))      # zero rows: masked keys, outside the pooling; their gradient is sliced off
    return _EncoderFn.apply(None, None, x, mhsa.W_Q.weight, mhsa.W_Q.bias, mhsa.W_K.weight, mhsa.W_K.bias,
                            mhsa.W_V.weight, mhsa.W_V.bias, additive.linear.weight, additive.linear.bias,
                            additive.attention_query_vector, S, 0.0, 0, N)


class _SplitRowsFn(torch.autograd.Function):
    """x[:n], x[n:] for the models' "one encoder pass over candidates + history" layout.  Plain slicing makes autograd build each part's
    gradient as a zero-filled full-size tensor with the slice copied in, and then add the two (5 kernels over 32 MB buffers per NRMS
    step); the backward here is one concatenation -- or none at all: the consumers of the two parts (the scorer for the candidates, the user
    encoder's first stage for the history) ask grad_dst() where their input gradient should go and write it into the two halves of ONE
    buffer, which the backward then returns as it is."""

    @staticmethod
    def forward(ctx, x, n):
        a, b = x[:n], x[n:]
        if x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and any(ctx.needs_input_grad):
            ctx.key = (a.data_ptr(), b.data_ptr())
            slot = {'shape': tuple(x.shape), 'n': n, 'buf': None, 'device': x.device, 'taken': [False, False]}
            _grad_dst[ctx.key[0]] = (slot, 0)
            _grad_dst[ctx.key[1]] = (slot, 1)
            while len(_grad_dst) > 16:                    # forwards that never ran a backward
                _grad_dst.pop(next(iter(_grad_dst)))
        else:
            ctx.key = None
        return a, b

    @staticmethod
    def backward(ctx, ga, gb):
        slot = None
        if ctx.key is not None:
            e = _grad_dst.pop(ctx.key[0], None)
            _grad_dst.pop(ctx.key[1], None)
            slot = e[0] if e is not None else None
        buf = slot['buf'] if slot is not None else None
        if buf is not None and ga is not None and gb is not None:
            n = slot['n']
            if (ga.data_ptr() == buf.data_ptr() and gb.data_ptr() == buf[n:].data_ptr() and ga.is_contiguous() and gb.is_contiguous()
                    and ga.dtype == buf.dtype and gb.dtype == buf.dtype and ga.numel() + gb.numel() == buf.numel()):
                return buf, None                           # both consumers wrote in place
        return torch.cat([ga, gb], dim=0), None


_grad_dst = collections.OrderedDict()          # data_ptr of a split_rows() part -> (slot, which part)


def grad_dst(x, shape=None):
    """Where the gradient with respect to ``x`` should be written, if ``x`` is (a view of) a part returned by split_rows(): a float32 view of
    the shared gradient buffer, shaped like ``x`` (or like ``shape``, same element count); None = allocate your own.  Looked up by address, so a
    reshaped view of the part finds it too.  Only valid during the backward pass of the step that produced ``x``."""
    e = _grad_dst.get(x.data_ptr())
    if e is None or not _GRAD_DST:
        return None
    slot, which = e
    n = slot['n']
    rows, cols = slot['shape']
    part_rows = n if which == 0 else rows - n
    if x.numel() != part_rows * cols or x.dtype != torch.float32 or not x.is_contiguous() or slot['taken'][which]:
        return None
    slot['taken'][which] = True                # each half is handed out once: a second asker (an unrelated tensor at a recycled address) allocates its own
    if slot['buf'] is None:
        slot['buf'] = torch.empty(rows, cols, dtype=torch.float32, device=slot['device'])
    part = slot['buf'][:n] if which == 0 else slot['buf'][n:]
    return part.view(shape if shape is not None else x.shape)


_GRAD_DST = os.environ.get('NR_GRAD_DST', '1') == '1'           # A/B knob: 0 = every consumer allocates its own input gradient (one concatenation more)


def split_rows(x, n):
    """(x[:n], x[n:]) with a single-kernel (or no-kernel) backward."""
    return _SplitRowsFn.apply(x, int(n))


# ----------------------------------------------------------------------------------------------------------
# dot-product scorer
# ----------------------------------------------------------------------------------------------------------
class _DotScoreFn(torch.autograd.Function):
    """src/model/general/click_predictor/dot_product.py:8-19."""

    @staticmethod
    def forward(ctx, cand, user):
        B, C, D = cand.shape
        c, u = _f32c(cand), _f32c(user)
        out = torch.empty(B, C, dtype=torch.float32, device=cand.device)
        _call('nr_score_dot', _lib().nr_score_dot, _ptr(c), _ptr(u), _ptr(out), B, C, D, _stream())
        ctx.save_for_backward(c, u)
        ctx.cand_in = cand if cand.data_ptr() == c.data_ptr() else None           # the caller's tensor: grad_dst() looks it up by address
        return out

    @staticmethod
    def backward(ctx, dl):
        c, u = ctx.saved_tensors
        B, C, D = c.shape
        dl = dl.to(torch.float32).contiguous()
        dc = grad_dst(ctx.cand_in) if ctx.cand_in is not None else None
        if dc is None:
            dc = torch.empty_like(c)
        du = torch.empty_like(u)
        _call('nr_score_dot_bwd', _lib().nr_score_dot_bwd, _ptr(dl), _ptr(c), _ptr(u), _ptr(dc), _ptr(du), B, C, D, _stream())
        return dc, du


class _DotScoreCEFn(torch.autograd.Function):
    """mean cross entropy of the dot-product click scores (dot_product.py:8-19 + train.py:205-206) in two launches forward, one backward
    (nr_score_ce_fwd / nr_score_ce_bwd) instead of scorer + log_softmax + nll_loss and their three backward kernels."""

    @staticmethod
    def forward(ctx, cand, user, target):
        B, C, D = cand.shape
        c, u = _f32c(cand), _f32c(user)
        dev = cand.device
        buf = torch.empty(2 * B * C + B + 1, dtype=torch.float32, device=dev)           # logits | dl | loss rows | loss
        logits, dl, rows, loss = buf[:B * C], buf[B * C:2 * B * C], buf[2 * B * C:2 * B * C + B], buf[2 * B * C + B:]
        _call('nr_score_ce_fwd', _lib().nr_score_ce_fwd, _ptr(c), _ptr(u), _ptr(target), _ptr(logits), _ptr(dl), _ptr(rows), _ptr(loss),
              B, C, D, _stream())
        ctx.save_for_backward(c, u, dl)
        ctx.cand_in = cand if cand.data_ptr() == c.data_ptr() else None           # the caller's tensor: grad_dst() looks it up by address
        ctx.logits = logits.view(B, C)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        c, u, dl = ctx.saved_tensors
        B, C, D = c.shape
        g = g.to(torch.float32).contiguous()
        dc = grad_dst(ctx.cand_in) if ctx.cand_in is not None else None
        if dc is None:
            dc = torch.empty_like(c)
        du = torch.empty_like(u)
        _call('nr_score_ce_bwd', _lib().nr_score_ce_bwd, _ptr(dl), _ptr(g), _ptr(c), _ptr(u), _ptr(dc), D, _ptr(du), D, B, C, D, _stream())
        return dc, du, None


def dot_score_ce(cand, user, target=None):
    """CrossEntropyLoss()(DotProductClickPredictor()(cand, user), target), mean over the batch; target None = class 0 for every impression (what
    the training loop builds with torch.zeros, train.py:205).  cand f32 [B, C, D], user f32 [B, D] on the GPU; target int64 [B] on the GPU."""
    _require_cuda(cand, "candidate_news_vector")
    B, C, D = cand.shape
    if D % 4 or not 1 <= C <= 64 or B < 1:
        raise NotImplementedError("dot_score_ce: feature dim must be a multiple of 4, 1..64 candidates per impression, a non-empty batch")
    if target is not None:
        _require_cuda(target, "target")
        if target.dtype != torch.int64 or tuple(target.shape) != (B,):
            raise ValueError("dot_score_ce: target must be int64 [B]")
        target = target.contiguous()
    return _DotScoreCEFn.apply(cand, user, target)


def dot_score(cand, user):
    _require_cuda(cand, "candidate_news_vector")
    if cand.shape[-1] % 4:
        raise NotImplementedError("feature dim must be a multiple of 4")
    return _DotScoreFn.apply(cand, user)


def score_csr(news_mat, user_mat, cand_idx, cand_ptr, user_idx):
    """Batched replacement of the per-impression loop of src/evaluate.py:245-260 (see nr_score_csr)."""
    _require_cuda(news_mat, "news matrix")
    nnz = int(cand_idx.numel())
    out = torch.empty(nnz, dtype=torch.float32, device=news_mat.device)
    n, u = _f32c(news_mat), _f32c(user_mat)
    _call('nr_score_csr', _lib().nr_score_csr, _ptr(n), _ptr(u), _ptr(cand_idx), _ptr(cand_ptr), _ptr(user_idx), _ptr(out),
                            int(user_idx.numel()), nnz, n.shape[1], _stream())
    return out


# ----------------------------------------------------------------------------------------------------------
# stand-alone L0 modules (same math, un-fused entry points for callers that use the primitives directly)
# ----------------------------------------------------------------------------------------------------------
class _MhsaFn(torch.autograd.Function):
    """MultiHeadSelfAttention.forward(Q, length=length) with K=V=Q (multihead_self.py:46-75) on a dense input already padded to an
    instantiated length; key_len int32 [n_seq] or None."""

    @staticmethod
    def forward(ctx, x, Wq, bq, Wk, bk, Wv, bv, key_len=None):
        lib = _lib()
        n_seq, S, _ = x.shape
        if not lib.nr_supported_seq_len(S):
            raise NotImplementedError(f"sequence length {S} is not instantiated in the HIP kernels (20, 50)")
        dev = x.device
        need_grad = any(ctx.needs_input_grad)
        Wp, bp = pack_qkv(Wq, bq, Wk, bk, Wv, bv)
        xd = _f32c(x)
        cbuf = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev)
        sp4 = (S + 3) // 4 * 4
        qs = ks = vts = None
        if need_grad:
            qs = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev)
            ks = torch.empty(n_seq * S, NR_KP, dtype=_BF16_AS_I16, device=dev)
            vts = torch.empty(n_seq, NR_HEADS, NR_DK, sp4, dtype=_BF16_AS_I16, device=dev)
        _call('nr_mhsa_fwd', lib.nr_mhsa_fwd_len, None, None, 0, _ptr(xd), _ptr(Wp), _ptr(bp), _ptr(cbuf), _ptr(qs), _ptr(ks), _ptr(vts), None,
                            _ptr(key_len), n_seq, S, 0.0, 0, _stream())
        if need_grad:
            ctx.save_for_backward(xd, qs, ks, vts, Wp, key_len)
            ctx.WdX = pack_qkv_dx(Wq, Wk, Wv)
        return _bf16(cbuf)[:, :NR_D].float().view(n_seq, S, NR_D)

    @staticmethod
    def backward(ctx, g):
        lib = _lib()
        xd, qs, ks, vts, Wp, key_len = ctx.saved_tensors
        n_seq, S, _ = xd.shape
        dev = xd.device
        ntok = n_seq * S
        dctx = g.reshape(ntok, NR_D).to(torch.bfloat16).contiguous()
        zw = torch.zeros(n_seq, S, dtype=torch.float32, device=dev)
        zg = torch.zeros(n_seq, NR_D, dtype=torch.float32, device=dev)
        dqkv = _workspace('dqkv', (ntok, NR_LDG), _BF16_AS_I16, dev, zero=True)
        _call('nr_attn_bwd', lib.nr_attn_bwd_len, _ptr(qs), _ptr(ks), _ptr(vts), _ptr(dctx), NR_D, _ptr(zw), _ptr(zg), _ptr(dqkv), _ptr(key_len),
              n_seq, S, 0.0, 0, _stream())
        Xb = _workspace('Xb', (ntok, NR_KP), _BF16_AS_I16, dev)
        _call('nr_gather_bf16', lib.nr_gather_bf16, None, None, 0, _ptr(xd), _ptr(Xb), ntok, 0.0, 0, _stream())
        # the engine's own GEMMs: split-K TN kernel for dW = dqkv^T [X | 1], nr_dx_gemm for dX = dqkv [Wq; Wk; Wv]
        dW_ext = sum_parts(gemm_tn_parts(dqkv, NR_LDG, Xb, NR_KP, 'nr_gemm_tn_dWqkv'))
        dXb = torch.empty(ntok, NR_KP, dtype=torch.bfloat16, device=dev)
        _call('nr_dx_gemm', lib.nr_dx_gemm, _ptr(dqkv), _ptr(ctx.WdX), _ptr(dXb), ntok, _stream())
        dX = dXb[:, :NR_D].float().view(n_seq, S, NR_D)
        gW = [dW_ext[i * NR_KP:i * NR_KP + NR_D, :NR_D] for i in range(3)]
        gb = [dW_ext[i * NR_KP:i * NR_KP + NR_D, NR_D] for i in range(3)]
        return dX, gW[0], gb[0], gW[1], gb[1], gW[2], gb[2], None


def mhsa_dense(x, mhsa, length=None):
    """MultiHeadSelfAttention.forward(Q, length=length), K = V = Q.  Any sequence length in [1, 50]: the input is zero-padded to an
    instantiated length and the padding is masked as keys; `length` (int tensor [batch], multihead_self.py:60-70) masks keys per
    sequence on top of that.  Rows of padded query positions are sliced off the result."""
    _require_cuda(x, "MultiHeadSelfAttention input")
    if not tuned_dims(x.shape[2], mhsa.num_attention_heads, None, x.shape[1]):
        from . import ops_generic
        x = x.to(torch.float32)
        return ops_generic.mhsa(x, x, x, mhsa, length)
    n_seq, L, _ = x.shape
    S = padded_len(L, "sequence length")
    key_len = None
    if length is not None:
        key_len = length.to(device=x.device, dtype=torch.int32).reshape(-1).clamp(max=L).contiguous()
        if key_len.numel() != n_seq:
            raise ValueError(f"length must have one entry per sequence ({n_seq}), got {key_len.numel()}")
    elif S != L:
        key_len = uniform_lengths(n_seq, L, x.device)
    if S != L:
        x = torch.nn.functional.pad(x, (0, 0, 0, S - L))
    y = _MhsaFn.apply(x, mhsa.W_Q.weight, mhsa.W_Q.bias, mhsa.W_K.weight, mhsa.W_K.bias, mhsa.W_V.weight, mhsa.W_V.bias, key_len)
    return y[:, :L] if S != L else y


class _AdditiveFn(torch.autograd.Function):
    """AdditiveAttention.forward (additive.py:27-53) on a dense [n_seq, S, D] input."""

    @staticmethod
    def forward(ctx, x, Wa, ba, qv, valid=None):
        lib = _lib()
        n_seq, S, _ = x.shape
        if not lib.nr_supported_seq_len(S):
            raise NotImplementedError(f"sequence length {S} is not instantiated in the HIP kernels (20, 50)")
        valid = S if valid is None else int(valid)
        dev = x.device
        Wap, bap, qvp = pack_additive(Wa, ba, qv)
        cb = torch.zeros(n_seq * S, NR_KP, dtype=torch.bfloat16, device=dev)
        cb[:, :NR_D] = x.detach().reshape(n_seq * S, NR_D).to(torch.bfloat16)
        cb[:, NR_D] = 1.0
        cbuf = cb.view(_BF16_AS_I16)
        out = torch.empty(n_seq, NR_D, dtype=torch.float32, device=dev)
        aw = torch.empty(n_seq, S, dtype=torch.float32, device=dev)
        _call('nr_additive_fwd', lib.nr_additive_fwd_v, _ptr(cbuf), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(out), NR_D, None, 0, _ptr(aw), n_seq, S,
              valid, _stream())
        ctx.save_for_backward(cbuf, aw, Wap, bap, qvp, out)
        ctx.qdim = Wa.shape[0]
        ctx.WaT = None if pool_flat_ok(S, False, n_seq, qdim=Wa.shape[0]) else pack_additive_t(Wa)
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = _lib()
        cbuf, aw, Wap, bap, qvp, y = ctx.saved_tensors
        n_seq, S = aw.shape
        dev = cbuf.device
        ntok = n_seq * S
        g_out = g_out.to(torch.float32).contiguous()
        qdim = ctx.qdim
        flat = pool_flat_ok(S, False, n_seq, qdim=qdim)
        if not flat:
            nwg = lib.nr_additive_bwd_grid(n_seq, S)
            dpre = _workspace('dpre', (ntok, NR_QP), _BF16_AS_I16, dev)
            dq_part = _workspace('dqp', (nwg, NR_QP), torch.float32, dev)
        # the fused backward (dctx = dpre @ Wa inside the kernel), the direct term added by nr_additive_dx, dWa by the TN kernel
        if flat:
            dpre, dq_part, dgemm = pool_bwd_flat(cbuf, Wap, bap, qvp, aw, g_out, _ptr(y), y.stride(0), n_seq, S, qdim, 'dense')
        else:
            dgemm = _workspace('dctx', (ntok, NR_KP), _BF16_AS_I16, dev)
            _call('nr_additive_bwd', lib.nr_additive_bwd_ex, _ptr(cbuf), _ptr(Wap), _ptr(bap), _ptr(qvp), _ptr(aw), _ptr(g_out), _ptr(dpre), _ptr(dq_part),
                  _ptr(ctx.WaT), _ptr(dgemm), n_seq, S, _stream())
        dx = torch.empty(n_seq, S, NR_D, dtype=torch.float32, device=dev)
        _call('nr_additive_dx', lib.nr_additive_dx, _ptr(dgemm), NR_KP, _ptr(aw), _ptr(g_out), _ptr(dx), n_seq, S, 0, _stream())
        dWa_ext = sum_parts(gemm_tn_parts(dpre, NR_QP, cbuf, NR_KP, 'nr_gemm_tn_dWa'))
        return dx, dWa_ext[:qdim, :NR_D], dWa_ext[:qdim, NR_D], dq_part.sum(dim=0)[:qdim], None


def additive_dense(x, additive, return_weights=False):
    """AdditiveAttention.forward on [batch, L, D], L in [1, 50]: zero-padded to an instantiated length and pooled over the first L.
    return_weights: also the attention weights [batch, L] (the tensorboard hook of additive.py:40-49 logs their batch mean) -- through the
    general-geometry kernels, which hand them out."""
    _require_cuda(x, "AdditiveAttention input")
    if return_weights or not tuned_dims(x.shape[2], NR_HEADS, additive.linear.weight.shape[0], x.shape[1]):
        from . import ops_generic
        return ops_generic.additive(x.to(torch.float32), additive, return_weights)
    L = x.shape[1]
    S = padded_len(L, "sequence length")
    if S != L:
        x = torch.nn.functional.pad(x, (0, 0, 0, S - L))
    return _AdditiveFn.apply(x, additive.linear.weight, additive.linear.bias, additive.attention_query_vector, L)
