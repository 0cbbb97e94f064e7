"""Optimiser step and token sort on CPU: the Adam oracle is pinned against torch.optim.Adam itself (the reference's optimiser,
src/train.py:127-128), then the engine's kernels -- compiled for the CPU wave emulator from the SAME sources (tests/emu) -- are held to
the oracle; ``EngineAdam`` (host logic: flat layout, lazy row-sparse tables, state_dict format) is driven end to end on CPU tensors
with the emulated library injected."""
import numpy as np
import pytest
import torch

from oracle import adam_numpy as onp
from tests.backends import EmuBackend

F = np.float32


@pytest.fixture(scope='module')
def emu():
    return EmuBackend()


def _sched(lr, betas, n):
    from news_recommendation_amd.optim import AdamSchedule
    return AdamSchedule.host_table(lr, betas, n)


def test_oracle_matches_torch_adam():
    """c3-style pinning: the numpy restatement vs torch.optim.Adam (the arithmetic the reference actually runs) over 8 steps."""
    rng = np.random.default_rng(0)
    p0 = rng.normal(size=(37, 11)).astype(F)
    grads = [rng.normal(size=p0.shape).astype(F) * F(10.0 ** rng.integers(-4, 1)) for _ in range(8)]
    grads[3] = None                                                  # a step without gradient signal (all-zero gradient)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-3)
    for g in grads:
        tp.grad = torch.zeros_like(tp) if g is None else torch.from_numpy(g.copy())
        opt.step()
    p, m, v = onp.adam_run(p0, grads, lr=1e-3)
    st = opt.state[tp]
    np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(m, st['exp_avg'].numpy(), rtol=1e-5, atol=1e-8)       # torch's lerp may fuse; a few ulp where g - m cancels
    np.testing.assert_allclose(v, st['exp_avg_sq'].numpy(), rtol=1e-5, atol=1e-12)


def test_schedule_table_is_torch_scalars():
    tab = _sched(1e-4, (0.9, 0.999), 50)
    for s in (1, 2, 7, 49):
        ss, bc = onp.scalars(1e-4, (0.9, 0.999), s)
        assert tab[s, 0] == ss and tab[s, 1] == bc


@pytest.mark.parametrize('n', [1, 3, 4, 1023, 4100])
def test_adam_flat_kernel_bit_exact(emu, n):
    rng = np.random.default_rng(n)
    p, g = rng.normal(size=n).astype(F), rng.normal(size=n).astype(F)
    m, v = (rng.normal(size=n) * 0.1).astype(F), (rng.random(size=n) * 0.01).astype(F)
    sched = np.ascontiguousarray(_sched(1e-4, (0.9, 0.999), 16))
    pe, me, ve = onp.adam_step(p, g, m, v, 5, grad_scale=0.5)
    # 16-byte aligned buffers
    bufs = [np.zeros(n + 8, dtype=F) for _ in range(4)]
    for b, src in zip(bufs, (p, g, m, v)):
        b[:n] = src
    assert emu.lib.nr_adam_flat(bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data, bufs[3].ctypes.data, n, sched.ctypes.data, 5,
                                0.9, 0.999, 1e-8, 0.5, 1, None) == 0
    assert np.array_equal(bufs[0][:n], pe) and np.array_equal(bufs[2][:n], me) and np.array_equal(bufs[3][:n], ve)
    assert not bufs[1][:n].any()                                      # gradient cleared in the same pass
    assert not any(b[n:].any() for b in bufs)                         # nothing written past n


def test_adam_flat_rejects_bad_args(emu):
    a = np.zeros(8, dtype=F)
    s = np.ascontiguousarray(_sched(1e-4, (0.9, 0.999), 4))
    assert emu.lib.nr_adam_flat(a.ctypes.data, a.ctypes.data, a.ctypes.data, a.ctypes.data, 8, s.ctypes.data, 0, 0.9, 0.999, 1e-8, 1.0, 1, None) != 0
    assert emu.lib.nr_adam_flat(None, a.ctypes.data, a.ctypes.data, a.ctypes.data, 8, s.ctypes.data, 1, 0.9, 0.999, 1e-8, 1.0, 1, None) != 0
    assert emu.lib.nr_adam_flat(a.ctypes.data, a.ctypes.data, a.ctypes.data, a.ctypes.data, 8, s.ctypes.data, 1, 1.5, 0.999, 1e-8, 1.0, 1, None) != 0


@pytest.mark.parametrize('with_counter', [False, True])
def test_row_lazy_adam_equals_dense_bitwise(emu, with_counter):
    """The lazy row kernels replay exactly what the dense kernel does to rows without gradient: same bits after 12 steps in which
    rows go idle for up to 11 steps, are re-read (catch-up), re-touched, duplicated within a step, and finally flushed.
    with_counter: the step index comes from a device step counter bumped at the START of every step (HIP-graph replays, graph.py) --
    the by-value step arguments are then frozen at their capture-time values (here: 1) and must be ignored."""
    rng = np.random.default_rng(1)
    ctr = np.zeros(1, dtype=np.uint32)
    if with_counter:
        assert emu.lib.nr_set_step_counter(ctr.ctypes.data) == 0
    try:
        _row_lazy_body(emu, rng, ctr, with_counter)
    finally:
        assert emu.lib.nr_set_step_counter(None) == 0


def _row_lazy_body(emu, rng, ctr, with_counter):
    R, d, T = 23, 130, 12
    p0 = rng.normal(size=(R, d)).astype(F)
    sched = np.ascontiguousarray(_sched(1e-3, (0.9, 0.999), T + 2))
    lib = emu.lib
    # dense reference: the flat kernel over the whole table every step
    pd, md, vd = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    pl, ml, vl = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    last = np.zeros(R, dtype=np.int32)
    for t in range(1, T + 1):
        nb = int(rng.integers(1, 6))
        ids = rng.integers(0, R, size=nb).astype(np.int64)
        if t == 4:
            ids[:] = ids[0]                                           # one row several times in a step
        rows = rng.normal(size=(nb, d)).astype(F)
        # forward would read these rows first: catch them up to t - 1
        read = np.unique(np.concatenate([ids, rng.integers(0, R, size=2)])).astype(np.int64)
        before = pl.copy()
        ctr[0] = t                                                    # the step's first node bumps the counter
        tv = 1 if with_counter else t                                 # by-value step index: frozen under a counter
        assert lib.nr_row_adam_catchup(read.ctypes.data, len(read), pl.ctypes.data, ml.ctypes.data, vl.ctypes.data, last.ctypes.data, R, d,
                                       sched.ctypes.data, tv - 1, 0.9, 0.999, 1e-8, None) == 0
        assert np.array_equal(pl[read], pd[read]), f"rows not current before the forward of step {t}"
        untouched = np.setdiff1d(np.arange(R), read)
        assert np.array_equal(pl[untouched], before[untouched])
        # dense: g = scatter-add of rows in position order, pad row 0 gets none
        g = np.zeros_like(p0)
        for i, r in zip(ids, rows):
            if i > 0:
                g[i] += r
        assert lib.nr_adam_flat(pd.ctypes.data, g.ctypes.data, md.ctypes.data, vd.ctypes.data, R * d, sched.ctypes.data, tv, 0.9, 0.999, 1e-8,
                                0.5, 1, None) == 0
        order = np.argsort(ids, kind='stable')
        ids_sorted, perm = ids[order].copy(), order.astype(np.int64)
        assert lib.nr_row_adam_step(ids_sorted.ctypes.data, perm.ctypes.data, nb, rows.ctypes.data, d,
                                    pl.ctypes.data, ml.ctypes.data, vl.ctypes.data, last.ctypes.data, R, d, sched.ctypes.data, tv,
                                    0.9, 0.999, 1e-8, 0.5, 0, None) == 0
    assert not np.array_equal(pl, pd)                                 # lazy table is stale somewhere before the flush ...
    # a forward OUTSIDE a step (validation between replays): the counter equals the number of completed steps there, so counter - 1 would
    # leave the rows one step behind -- nr_row_adam_catchup_ex(by_value = 1) takes the index from the argument, counter attached or not
    stale = np.flatnonzero((last > 0) & (last < T))
    if len(stale):
        read = stale[:3].astype(np.int64)
        assert lib.nr_row_adam_catchup_ex(read.ctypes.data, len(read), pl.ctypes.data, ml.ctypes.data, vl.ctypes.data, last.ctypes.data, R, d,
                                          sched.ctypes.data, T, 1, 0.9, 0.999, 1e-8, None) == 0
        assert np.array_equal(pl[read], pd[read]) and (last[read] == T).all()
    assert lib.nr_row_adam_flush(pl.ctypes.data, ml.ctypes.data, vl.ctypes.data, last.ctypes.data, R, d, sched.ctypes.data, T,
                                 0.9, 0.999, 1e-8, None) == 0
    assert np.array_equal(pl, pd) and np.array_equal(ml, md) and np.array_equal(vl, vd)      # ... and bit-identical after it
    assert np.array_equal(pl[0], p0[0])                               # padding row never moves


@pytest.mark.parametrize('n,num_rows', [(1, 2), (777, 50), (5000, 70976), (4097, 130001), (9000, 400001), (2048, 1 << 20)])
def test_sort_ids_matches_stable_argsort(emu, n, num_rows):
    rng = np.random.default_rng(n)
    ids = rng.integers(0, num_rows, size=n).astype(np.int64)
    ids[rng.random(n) < 0.45] = 0                                     # the padding token dominates real token streams
    if n > 10:
        ids[3], ids[7] = -5, num_rows + 9                             # out-of-table ids are clamped like the forward gather
    ws_bytes = emu.lib.nr_sort_ids_workspace(n, num_rows)
    assert ws_bytes > 0
    ws = np.zeros(ws_bytes // 8 + 2, dtype=np.int64)
    out_ids, out_perm = np.full(n, -1, dtype=np.int64), np.full(n, -1, dtype=np.int64)
    assert emu.lib.nr_sort_ids(ids.ctypes.data, n, num_rows, out_ids.ctypes.data, out_perm.ctypes.data, ws.ctypes.data, ws_bytes, None) == 0
    clamped = np.clip(ids, 0, num_rows - 1)
    order = np.argsort(clamped, kind='stable')
    assert np.array_equal(out_perm, order)
    assert np.array_equal(out_ids, clamped[order])


def test_sort_ids_bad_args(emu):
    ids = np.zeros(4, dtype=np.int64)
    assert emu.lib.nr_sort_ids_workspace(4, 0) == -1
    assert emu.lib.nr_sort_ids(ids.ctypes.data, 4, 100, ids.ctypes.data, ids.ctypes.data, None, 0, None) != 0


# ---- EngineAdam host logic on CPU tensors (emulated kernels injected) -----------------------------------------------------------------
class _Toy(torch.nn.Module):
    def __init__(self, users=40, d=24):
        super().__init__()
        self.user_embedding = torch.nn.Embedding(users, d, padding_idx=0)
        self.lin = torch.nn.Linear(d, 3)


class _RowsFn(torch.autograd.Function):
    """Stand-in for ops_gru._UserRowsFn on CPU: same protocol towards the optimiser (_nr_row_sync before the read, _nr_row_sink in the
    backward, otherwise a dense scatter into ops.grad_target)."""

    @staticmethod
    def forward(ctx, ids, table):
        sync = getattr(table, '_nr_row_sync', None)
        if sync is not None:
            sync(ids)
        ctx.save_for_backward(ids)
        ctx.table = table
        return table.detach()[ids].clone()

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        sink = getattr(ctx.table, '_nr_row_sink', None)
        if sink is not None:
            sink(ids, g)
            return None, None
        d = torch.zeros_like(ctx.table)
        keep = ids != 0
        d.index_add_(0, ids[keep], g[keep])
        return None, d


def _toy_loss(model, ids, y):
    return torch.nn.functional.cross_entropy(model.lin(_RowsFn.apply(ids, model.user_embedding.weight)), y)


def test_engine_adam_matches_torch_adam_and_state_dict_format(emu):
    from news_recommendation_amd.optim import EngineAdam
    torch.manual_seed(0)
    a, b = _Toy(), _Toy()
    b.load_state_dict(a.state_dict())
    ref_opt = torch.optim.Adam(a.parameters(), lr=1e-2)
    opt = EngineAdam(b, lr=1e-2, row_sparse=('user_embedding.weight',), lib=emu.lib, stream_fn=lambda: None)
    assert b.user_embedding.weight.grad is None and b.lin.weight.grad.data_ptr() == opt.flat_g.data_ptr() + opt.slices['lin.weight'][0] * 4
    g = torch.Generator().manual_seed(1)
    for step in range(9):
        ids = torch.randint(0, 40, (6,), generator=g)
        y = torch.randint(0, 3, (6,), generator=g)
        ref_opt.zero_grad()
        _toy_loss(a, ids, y).backward()
        ref_opt.step()
        opt.zero_grad()
        _toy_loss(b, ids, y).backward()
        opt.step()
        np.testing.assert_allclose(b.lin.weight.detach().numpy(), a.lin.weight.detach().numpy(), rtol=1e-5, atol=1e-7)
        assert not opt.flat_g.any()
    sd = b.state_dict()                                               # pre-hook flushes the lazy table
    np.testing.assert_allclose(sd['user_embedding.weight'].numpy(), a.user_embedding.weight.detach().numpy(), rtol=1e-5, atol=1e-7)
    # the optimiser state loads into torch.optim.Adam and vice versa (src/train.py:151-152,268-275)
    osd = opt.state_dict()
    probe = torch.optim.Adam(_Toy().parameters(), lr=1e-2)
    probe.load_state_dict(osd)
    ref_sd = ref_opt.state_dict()
    assert set(osd['state']) == set(ref_sd['state']) and osd['param_groups'][0]['params'] == ref_sd['param_groups'][0]['params']
    for i in ref_sd['state']:
        assert float(osd['state'][i]['step']) == float(ref_sd['state'][i]['step'])
        np.testing.assert_allclose(osd['state'][i]['exp_avg'].numpy(), ref_sd['state'][i]['exp_avg'].numpy(), rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(osd['state'][i]['exp_avg_sq'].numpy(), ref_sd['state'][i]['exp_avg_sq'].numpy(), rtol=1e-5, atol=1e-12)
    # resume from torch's checkpoint: both continue in lock-step
    c = _Toy()
    c.load_state_dict({k: v.clone() for k, v in a.state_dict().items()})
    opt_c = EngineAdam(c, lr=1e-2, row_sparse=('user_embedding.weight',), lib=emu.lib, stream_fn=lambda: None)
    opt_c.load_state_dict(ref_sd)
    assert opt_c.t == 9
    for step in range(3):
        ids = torch.randint(0, 40, (6,), generator=g)
        y = torch.randint(0, 3, (6,), generator=g)
        ref_opt.zero_grad()
        _toy_loss(a, ids, y).backward()
        ref_opt.step()
        _toy_loss(c, ids, y).backward()
        opt_c.step()
    np.testing.assert_allclose(c.state_dict()['user_embedding.weight'].numpy(), a.user_embedding.weight.detach().numpy(), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(c.lin.bias.detach().numpy(), a.lin.bias.detach().numpy(), rtol=2e-5, atol=1e-7)


def test_fault_words_gate_every_optimiser_kernel_and_the_steps_are_repeated(emu):
    """A persistent GRU sweep that gives up a wait sets STICKY bits in the process's fault words (csrc/k_xcd.h); from then on no optimiser kernel
    applies an update (csrc/k_optim.h) until the host has looked.  Here the bits are set by hand in the middle of a run: the gated steps change
    nothing but the (cleared) gradient buffer, the first skipped step index is recorded, a clean sweep afterwards does NOT erase the evidence
    (the per-launch error word was zeroed by the next launch), and after rewind_after_fault() the repeated steps reproduce the
    undisturbed run bit for bit."""
    from news_recommendation_amd.optim import EngineAdam
    torch.manual_seed(0)
    a, b = _Toy(), _Toy()
    b.load_state_dict(a.state_dict())
    mk = lambda m: EngineAdam(m, lr=1e-2, row_sparse=('user_embedding.weight',), lib=emu.lib, stream_fn=lambda: None)
    ref_opt, opt = mk(a), mk(b)
    words = opt.attach_fault_words()
    try:
        g = torch.Generator().manual_seed(3)
        batches = [(torch.randint(0, 40, (6,), generator=g), torch.randint(0, 3, (6,), generator=g)) for _ in range(9)]
        for ids, y in batches:                                       # the undisturbed run (its kernels see the attached words too: all zero)
            _toy_loss(a, ids, y).backward()
            ref_opt.step()
        assert words.tolist() == [0, 0, 0, 0]
        snap = None
        for k, (ids, y) in enumerate(batches):
            if k == 4:                                               # the sweep of step 5 fails (backward sweep, "a wait gave up")
                words[1] = 2
                snap = ({n: p.detach().clone() for n, p in b.named_parameters()}, opt.flat_m.clone(), opt.flat_v.clone(),
                        opt.sparse[0].m.clone(), opt.sparse[0].last.clone())
            _toy_loss(b, ids, y).backward()
            opt.step()
            if k >= 4:
                assert not opt.flat_g.any()                          # the garbage gradient is dropped, not kept for the next step
        assert words.tolist() == [0, 2, 5, 0]                        # sticky bits + the first skipped step; later (clean) steps erase nothing
        for n, p in b.named_parameters():
            assert torch.equal(p.detach(), snap[0][n]), n            # nothing moved since step 4: dense ...
        assert torch.equal(opt.flat_m, snap[1]) and torch.equal(opt.flat_v, snap[2])
        assert torch.equal(opt.sparse[0].m, snap[3]) and torch.equal(opt.sparse[0].last, snap[4])       # ... and row-sparse (catch-up included)
        fw, bw = ctypes_words(emu)
        assert (fw, bw) == (0, 2)
        first = opt.rewind_after_fault()
        assert first == 5 and opt.t == 4 and words.tolist() == [0, 0, 0, 0] and opt.rewind_after_fault() is None
        for ids, y in batches[first - 1:]:
            _toy_loss(b, ids, y).backward()
            opt.step()
        assert opt.t == ref_opt.t == 9
        sa, sb = a.state_dict(), b.state_dict()
        for n in sa:
            assert torch.equal(sa[n], sb[n]), n
        assert torch.equal(opt.flat_m, ref_opt.flat_m) and torch.equal(opt.sparse[0].v, ref_opt.sparse[0].v)
    finally:
        opt.detach_fault_words()
    w4 = (__import__('ctypes').c_uint32 * 4)()
    assert emu.lib.nr_fault_state(w4) == 0 and list(w4) == [0, 0, 0, 0]        # back on the library's own block


def ctypes_words(emu):
    import ctypes
    f, b = ctypes.c_int32(-1), ctypes.c_int32(-1)
    assert emu.lib.nr_gru_persist_status(ctypes.byref(f), ctypes.byref(b)) == 0
    return f.value, b.value
