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
