"""Data-parallel gradient exchange on CPU (gloo, world_size 2): the same FlatGradBuffer / hook code that runs over
RCCL on the GPUs.  Each rank computes gradients of a toy model on its own shard; after the all-reduce every rank
must hold the mean gradient, and one optimizer step must keep the replicas bit-identical."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from news_recommendation_amd import dist as nrdist
    r, w, _ = nrdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # different init on purpose
    model = torch.nn.Sequential(torch.nn.Embedding(50, 8, padding_idx=0), torch.nn.Flatten(), torch.nn.Linear(32, 3))
    nrdist.broadcast_parameters(model)                  # -> identical replicas
    ref = [p.detach().clone() for p in model.parameters()]
    torch.manual_seed(7 + rank)
    x = torch.randint(0, 50, (6, 4))
    y = torch.randint(0, 3, (6,))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    if mode == 'flat':
        fgb = nrdist.FlatGradBuffer(model.parameters())
        fgb.zero()
        torch.nn.functional.cross_entropy(model(x), y).backward()
        assert fgb.check_views()
        local = fgb.flat.clone()
        fgb.allreduce_mean()
        got = fgb.flat.clone()
    else:
        nrdist.attach_grad_allreduce(model)             # unmodified loop: zero_grad / backward / step
        opt.zero_grad()
        torch.nn.functional.cross_entropy(model(x), y).backward()
        got = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        local = None
    opt.step()
    q.put((rank, [t.numpy() for t in ref], None if local is None else local.numpy(), got.numpy(),
           [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['flat', 'hook'])
def test_grad_allreduce_world2(mode):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    (_, ref0, loc0, got0, new0), (_, ref1, loc1, got1, new1) = res
    for a, b in zip(ref0, ref1):
        assert np.array_equal(a, b)                     # broadcast made the replicas identical
    assert np.array_equal(got0, got1)                   # both ranks hold the same reduced gradient
    if mode == 'flat':
        np.testing.assert_allclose(got0, (loc0 + loc1) / 2, rtol=1e-6, atol=1e-8)
        assert not np.array_equal(loc0, loc1)
    for a, b in zip(new0, new1):
        assert np.array_equal(a, b)                     # replicas stay in lock-step after the optimizer step


# ---- EngineAdam over gloo: bucketed all-reduce started from the backward + touched-row exchange --------------------------------------------
class _TableFn(torch.autograd.Function):
    """CPU stand-in for the engine's embedding backward protocol (ops._EncoderFn): the table gradient is ACCUMULATED IN PLACE into
    ops.grad_target(param) (the flat-buffer view, nothing returned to autograd) and ops.table_grad_ready(param) is signalled as soon as it
    is complete -- before the remaining (weight) gradients are produced."""

    @staticmethod
    def forward(ctx, ids, table):
        ctx.save_for_backward(ids)
        ctx.table = table
        return table.detach()[ids].clone()

    @staticmethod
    def backward(ctx, g):
        from news_recommendation_amd import ops
        (ids,) = ctx.saved_tensors
        dst, ret = ops.grad_target(ctx.table)
        keep = ids != 0
        dst.index_add_(0, ids[keep], g[keep])
        ops.table_grad_ready(ctx.table)
        return None, ret


class _RowsFn(torch.autograd.Function):
    """CPU stand-in for ops_gru._UserRowsFn (row-sparse table: sync before the read, (ids, rows) to the sink in the backward)."""

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


class _ToyRec(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.word_embedding = torch.nn.Embedding(60, 8, padding_idx=0)
        self.user_embedding = torch.nn.Embedding(30, 8, padding_idx=0)
        self.lin = torch.nn.Linear(8, 3)

    def forward(self, words, users):
        w = _TableFn.apply(words, self.word_embedding.weight).sum(dim=1)
        u = _RowsFn.apply(users, self.user_embedding.weight)
        return self.lin(w + u)


def _toy_batches(rank, steps, ragged=False):
    """ragged: rank 1's batches 1 and 2 are shorter (3 and 1 samples instead of 5): fewer gradient rows than its peer in those steps."""
    g = torch.Generator().manual_seed(50 + rank)
    out = []
    for i in range(steps):
        n = (3 if i == 1 else 1) if (ragged and rank == 1 and i in (1, 2)) else 5
        out.append((torch.randint(0, 60, (n, 4), generator=g), torch.randint(0, 30, (n,), generator=g), torch.randint(0, 3, (n,), generator=g)))
    return out


def _engine_worker(rank, world, port, overlap, q, table_rs=False, ragged=False, segmented=False):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from news_recommendation_amd import dist as nrdist, optim
    from tests.backends import EmuBackend
    nrdist.init_from_env(backend='gloo')
    optim.TABLE_MIN_NUMEL = 256                         # the toy word table (480 elements) is the "table bucket"
    torch.manual_seed(100 + rank)
    model = _ToyRec()
    nrdist.broadcast_parameters(model)
    opt = optim.EngineAdam(model, lr=1e-2, row_sparse=('user_embedding.weight',), overlap=overlap, lib=EmuBackend().lib, stream_fn=lambda: None,
                           table_rs=table_rs)
    assert [r.name for r in opt.regions] == ['small', 'word_embedding.weight']
    assert (opt.regions[1].end - opt.regions[1].lo) % (world * 64) == 0 and opt.regions[1].hi - opt.regions[1].lo == 480
    if segmented:                               # the protocol of graph.SegmentedStep, issued by hand (the graphs themselves need a GPU)
        opt.overlap = False
        opt.set_row_capacity(5)
        opt.prepare_segments()
    for words, users, y in _toy_batches(rank, 4, ragged):
        torch.nn.functional.cross_entropy(model(words, users), y).backward()
        if segmented == 'overlap':              # the three-segment form: the table exchange in flight under the weight-gradient segment
            opt.begin_step()
            works = opt.start_tables()          # (behind graph A, which ends with the embedding scatter)
            opt.stage_rows()                    # (graph W: postponed weight-gradient GEMMs + row staging)
            opt.exchange_all(works)             # rows, small bucket, then wait for everything
            opt.apply_all()                     # (graph B: Adam -- under table_rs on this rank's shard)
            opt.gather_tables()                 # table_rs: collect the updated table
        elif segmented:
            opt.stage_rows()                    # (tail of graph A)
            opt.begin_step()
            opt.exchange_all()                  # the collectives between the segments
            opt.apply_all()                     # (graph B)
        else:
            opt.step()
        assert not opt.flat_g.any()
    assert opt.comm_bytes['small'] > 0 and opt.comm_bytes['word_embedding.weight'] >= 480 * 4 and opt.comm_bytes['user_embedding.weight'] == world * 5 * (8 + 32)
    sd = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    if table_rs:                                # a rank-local state_dict() must not start collectives on its own
        try:
            opt.state_dict()
            raise AssertionError('state_dict() of sharded moments did not ask for gather_state()')
        except RuntimeError as e:
            assert 'gather_state' in str(e)
    opt.gather_state()                          # collective under table_rs: the moment shards are gathered on every rank ...
    osd = opt.state_dict()                      # ... so that this is local (and raises if the gather was skipped)
    for i, st in osd['state'].items():
        sd[f'opt/{i}/exp_avg'] = st['exp_avg'].numpy().copy()
        sd[f'opt/{i}/exp_avg_sq'] = st['exp_avg_sq'].numpy().copy()
    if ragged and rank == 1:
        # a batch LONGER than the capacity the ranks agreed on at the first exchange is an error on the rank that sees it, raised before
        # any collective is entered (its peers are not left in a mismatched collective by this rank picking another protocol)
        st = opt.sparse[0]
        st.pending.append((torch.arange(1, 8), torch.ones(7, 8)))
        with pytest.raises(RuntimeError, match='agreed on at most 5'):
            opt._exchange_rows(st)
        st.pending.clear()
    q.put((rank, sd))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # what bench.py does after its timed region: every rank has left the group, rank 0 goes on alone -- nothing below may reach for
        # a collective (the optimiser asks for the world size at every step)
        assert optim._world() == 1
        words, users, y = _toy_batches(7, 1)[0]
        torch.nn.functional.cross_entropy(model(words, users), y).backward()
        opt.step()
        assert all(torch.isfinite(p).all() for p in model.parameters()) and not opt.flat_g.any()


@pytest.mark.parametrize('overlap', [True, False])
def test_engine_adam_world2_equals_single_process_mean_gradient(overlap):
    """Two ranks, each its own shard: table all-reduce (started inside the backward when overlap=True), small-bucket all-reduce, and the
    (row id, gradient row) all-gather for the row-sparse table must leave BOTH replicas bit-identical and equal to one process running
    torch.optim.Adam on the mean of the two shards' gradients -- i.e. sparse exchange == dense all-reduce."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import numpy as np
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), f"replicas diverged in {k}"
    _assert_equals_single_process(res[0])


def _assert_equals_single_process(got, ragged=False):
    """single-process reference: same initial weights (rank 0's init), dense torch Adam on the mean loss of both shards"""
    import numpy as np
    torch.manual_seed(100)
    ref = _ToyRec()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
    b0, b1 = _toy_batches(0, 4, ragged), _toy_batches(1, 4, ragged)
    for (w0, u0, y0), (w1, u1, y1) in zip(b0, b1):
        opt.zero_grad()
        loss = 0.5 * (torch.nn.functional.cross_entropy(ref(w0, u0), y0) + torch.nn.functional.cross_entropy(ref(w1, u1), y1))
        loss.backward()
        opt.step()
    for k, v in ref.state_dict().items():
        np.testing.assert_allclose(got[k], v.numpy(), rtol=2e-5, atol=1e-7, err_msg=k)


def _run_world2(overlap, table_rs=False, ragged=False, segmented=False):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, overlap, q, table_rs, ragged, segmented)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_engine_adam_table_reduce_scatter_equals_allreduce():
    """table_rs: reduce-scatter -> Adam on the rank's shard -> all-gather of the updated table leaves the replicas bit-identical to each
    other AND to the all-reduce form, parameters and (gathered) Adam moments alike (same bytes on the wire, 1/world of the
    update pass per GPU)."""
    import numpy as np
    rs, ar = _run_world2(True, table_rs=True), _run_world2(True, table_rs=False)
    assert any(k.startswith('opt/') for k in rs[0])
    for k in ar[0]:
        assert np.array_equal(rs[0][k], rs[1][k]), f"replicas diverged in {k}"
        assert np.array_equal(rs[0][k], ar[0][k]), f"reduce-scatter form differs from the all-reduce form in {k}"


def test_engine_adam_unequal_row_counts_keep_one_protocol():
    """A rank with a shorter batch pads its (row id, gradient row) message to the agreed capacity (id 0 = the padding row, skipped by the
    update) instead of switching collectives on its own: replicas stay identical and equal to the single-process mean-of-means update."""
    import numpy as np
    res = _run_world2(True, ragged=True)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), f"replicas diverged in {k}"
    _assert_equals_single_process(res[0], ragged=True)


@pytest.mark.parametrize('ragged', [False, True])
def test_engine_adam_segmented_step_equals_step(ragged):
    """The segmented form of the data-parallel step (graph.SegmentedStep: [backward + row staging] | collectives | [update]) leaves the
    replicas bit-identical to each other AND to EngineAdam.step() -- parameters and Adam moments, also when a rank's batch is shorter than the
    agreed row capacity (its send buffer is padded with the padding row)."""
    import numpy as np
    seg, ref = _run_world2(False, ragged=ragged, segmented=True), _run_world2(True, ragged=ragged)
    assert any(k.startswith('opt/') for k in seg[0])
    for k in ref[0]:
        assert np.array_equal(seg[0][k], seg[1][k]), f"replicas diverged in {k}"
        assert np.array_equal(seg[0][k], ref[0][k]), f"segmented step differs from step() in {k}"


@pytest.mark.parametrize('table_rs', [False, True])
def test_engine_adam_overlapped_segments_equal_step(table_rs):
    """the three-segment issue mode of graph.SegmentedStep (table exchange started behind the scatter segment, weight
    gradients + row staging under it, then rows / small bucket, Adam, and -- reduce-scatter form -- the gather of the updated table) leaves the
    replicas bit-identical to each other AND to EngineAdam.step(), parameters and Adam moments, in the all-reduce and the reduce-scatter form
    of the table bucket (which the two-segment mode of round 4 refused)."""
    import numpy as np
    seg, ref = _run_world2(False, table_rs=table_rs, segmented='overlap'), _run_world2(True, table_rs=table_rs)
    assert any(k.startswith('opt/') for k in seg[0])
    for k in ref[0]:
        assert np.array_equal(seg[0][k], seg[1][k]), f"replicas diverged in {k}"
        assert np.array_equal(seg[0][k], ref[0][k]), f"overlapped segments differ from step() in {k}"
