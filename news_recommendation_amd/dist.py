"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no multi-device path; impressions are independent, so the engine shards
them across ranks and the only exchange step is the gradient all-reduce before the optimizer step
(SURVEY.md 8 e1-e3).  All gradients are packed into ONE flat fp32 buffer (NRMS: 21.96 M params = 87.8 MB, 97 %
of it the embedding table) so that a single large collective runs per step: xGMI is a point-to-point mesh
(7 links x ~153 GB/s per GPU) and RCCL spreads one big message over all links, whereas many small
messages are latency-bound.

``backend='nccl'`` IS RCCL on ROCm; CPU tests use ``gloo`` with the same code path.
"""
import os
import torch
import torch.distributed as dist


def local_device_index(local_rank=None):
    """Index of this rank's GPU among the devices the process can SEE.  A launcher that narrows visibility to one GPU per rank
    (``HIP_VISIBLE_DEVICES=<LOCAL_RANK>``, so that the reference's hard-coded ``cuda:0`` is the local GPU) leaves exactly one
    visible device: the index is then 0, not LOCAL_RANK (``set_device(LOCAL_RANK)`` would be an invalid ordinal on every rank >= 1)."""
    if local_rank is None:
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n == 0:
        return 0
    if local_rank >= n:
        if n == 1:
            return 0
        raise RuntimeError(f"LOCAL_RANK={local_rank} but only {n} GPUs are visible (HIP_VISIBLE_DEVICES="
                           f"{os.environ.get('HIP_VISIBLE_DEVICES')!r}): expose one GPU per rank or all of them")
    return local_rank


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world, local device index)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = local_device_index()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # NR_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses duplicate devices) -- how the N > 1 code path of bench.py and
            # the optimiser's exchange are exercised on a single-GPU box (tools/gpu_two_ranks_one_gpu.sh); never a production setting
            backend = os.environ.get('NR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_parameters(model, src=0):
    """Identical initial weights on every rank."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=src)
        for b in model.buffers():
            dist.broadcast(b.data, src=src)
        from . import ops
        ops.invalidate_packed()                  # .data writes do not bump the parameters' version counters


class FlatGradBuffer:
    """All parameter gradients as views into one contiguous fp32 buffer.

    * ``zero()`` replaces ``optimizer.zero_grad()`` (keeps the views alive),
    * the parameters are marked ``_nr_inplace_grad``: the engine's embedding-table backward then accumulates into the view directly
      instead of returning a dense table-sized tensor for autograd to add (only valid with this explicit zero / all-reduce protocol:
      no autograd hook fires for a gradient that is not returned),
    * ``allreduce_mean()`` is the single collective of the step (sum then divide by world size).
    """

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            p._nr_inplace_grad = True        # ops.grad_target(): the table scatters accumulate straight into these views
            off += n
        self.nbytes = total * 4

    def zero(self):
        self.flat.zero_()

    def check_views(self):
        """autograd accumulates in place into an existing .grad, so the views must still alias the flat buffer."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * 4:
                return False
            off += p.numel()
        return True

    def allreduce_mean(self):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())


def allreduce_grads_mean(model):
    """Generic path for an unmodified training loop (optimizer.zero_grad() drops .grad tensors, so gradients are
    re-flattened every step): flatten -> one all-reduce -> copy back."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch._utils._flatten_dense_tensors(grads)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    for g, f in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
        g.copy_(f)


def attach_grad_allreduce(model):
    """Make an unmodified ``loss.backward(); optimizer.step()`` loop (src/train.py:231-233) data-parallel: once every
    parameter has accumulated its gradient, all-reduce them all (post-accumulate-grad hooks, no DDP wrapper)."""
    params = [p for p in model.parameters() if p.requires_grad]
    state = {'left': len(params)}

    def hook(_p):
        state['left'] -= 1
        if state['left'] == 0:
            state['left'] = len(params)
            allreduce_grads_mean(model)

    return [p.register_post_accumulate_grad_hook(hook) for p in params]
