"""The optimiser step of src/train.py:127-128,227-233 -- ``Adam(model.parameters(), lr)``, ``zero_grad / backward / step`` -- fused with the
data-parallel gradient exchange.

What the reference does per step: ``optimizer.zero_grad()``, ``loss.backward()`` (53 dense 85 MB embedding gradients accumulated),
``optimizer.step()`` (dense Adam over every parameter, one process, one device).  ``EngineAdam`` keeps exactly that update rule
(torch.optim.Adam defaults) and the ``state_dict`` format of ``torch.optim.Adam`` (checkpoints stay interchangeable with the reference's
``train.py:144-159,264-277``), on a memory layout made for the engine:

* all DENSE parameters live in one flat fp32 buffer ``[small parameters | word-embedding table(s)]``; ``.grad`` of every parameter is a
  view into a flat gradient buffer of the same layout, Adam moments likewise.  One kernel pass (``nr_adam_flat``) reads p, g, m, v,
  writes p, m, v and clears g: the zero_grad pass, the ``/ world`` pass and the separate moment updates are gone.
* data parallel (one process per GPU, RCCL over xGMI): the table bucket's all-reduce is started from inside the backward, right after
  the embedding scatter has been enqueued (``param._nr_grad_ready``), and runs on RCCL's stream while the encoder's weight-gradient
  GEMMs still compute; the small bucket (2.65 MB for NRMS) follows at the end.  xGMI is point-to-point (7 links per GPU): two large
  messages, not one per parameter.
* ROW-SPARSE tables (LSTUR ``user_embedding``, src/model/LSTUR/__init__.py:38-42: 711,223 x 900 fp32 = 2.56 GB at MIND-large scale, B
  rows touched per step) never materialise a dense gradient: the backward hands over ``(row ids, gradient rows)``, ranks all-gather
  those (B x 3.6 KB each instead of a 2.56 GB all-reduce), and ``nr_row_adam_*`` evaluate the dense Adam recurrence lazily and exactly
  (csrc/k_optim.h): rows are caught up just before the forward reads them, and ``flush()`` brings the whole table up to date before
  ``state_dict()`` / checkpoints.

There is no CPU implementation: the update runs in the HIP library (``_capi.load()``).  ``lib=`` lets the GPU-less unit tests inject
a host build of the same kernel sources that the test suite owns (test infrastructure; the product never loads it).
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _capi, ops

TABLE_MIN_NUMEL = 1 << 22          # dense parameters from 4 M elements up form their own all-reduce bucket (the embedding tables)
_ALIGN = 64                        # flat regions start on 256-byte boundaries


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class AdamSchedule:
    """Device table of the per-step scalars, computed in double exactly as torch's ``_single_tensor_adam`` does:
    ``sched[2 s] = lr / (1 - beta1 ** s)``, ``sched[2 s + 1] = sqrt(1 - beta2 ** s)``."""

    def __init__(self, lr, betas, device, capacity=4096):
        self.lr, self.betas, self.device = float(lr), (float(betas[0]), float(betas[1])), device
        self.capacity = 0
        self.table = None
        self.ensure(capacity - 1)

    @staticmethod
    def host_table(lr, betas, n):
        s = np.arange(n, dtype=np.float64)
        out = np.zeros((n, 2), dtype=np.float32)
        if n > 1:
            bc1 = 1.0 - np.power(betas[0], s[1:])
            bc2 = 1.0 - np.power(betas[1], s[1:])
            out[1:, 0] = (lr / bc1).astype(np.float32)
            out[1:, 1] = np.sqrt(bc2).astype(np.float32)
        return out

    def ensure(self, step):
        if step < self.capacity:
            return
        cap = max(4096, self.capacity)
        while cap <= step:
            cap *= 2
        self.table = torch.from_numpy(self.host_table(self.lr, self.betas, cap)).to(self.device)
        self.capacity = cap


class _Region:
    """[lo, hi): the parameters; [hi, end): zero padding so that a table region divides into `world` equal, 256-byte aligned shards."""
    __slots__ = ('name', 'lo', 'hi', 'end', 'work')

    def __init__(self, name, lo, hi, end=None):
        self.name, self.lo, self.hi, self.work = name, lo, hi, None
        self.end = hi if end is None else end


class _SparseTable:
    __slots__ = ('name', 'param', 'm', 'v', 'last', 'pending', 'pad_row', 'send_ids', 'send_rows', 'recv_ids', 'recv_rows')


class EngineAdam:
    """Adam over a model's parameters with the engine's flat layout, fused update kernel, bucketed / overlapped all-reduce and
    row-sparse tables.  Drop-in for the ``optimizer`` object of src/train.py: ``zero_grad()``, ``step()``, ``state_dict()``,
    ``load_state_dict()``, ``param_groups``."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, row_sparse=(), overlap=True, lib=None, stream_fn=None,
                 table_rs=None, force_dist=False):
        """table_rs (default: env NR_TABLE_RS=1): the table buckets are exchanged as reduce-scatter -> Adam on this rank's 1/world shard ->
        all-gather of the updated parameters instead of all-reduce -> Adam on the whole table (same bytes on the wire, 1/world of the
        update pass per GPU).  force_dist: take the multi-rank code path even in a world of one (the RCCL path can then be
        exercised on a single GPU: collectives over one rank are identities)."""
        if isinstance(model, torch.nn.Module):
            named = list(model.named_parameters())
            self._module = model
        else:
            named = list(model)
            self._module = None
        named = [(n, p) for n, p in named if p.requires_grad]
        if not named:
            raise ValueError("EngineAdam: no trainable parameters")
        self.lib = lib if lib is not None else _capi.load()
        self._stream_fn = stream_fn if stream_fn is not None else (lambda: torch.cuda.current_stream().cuda_stream)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.overlap = bool(overlap)
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]               # model.parameters() order == torch.optim.Adam's parameter indices
        dev = self.params[0].device
        self.device = dev
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("EngineAdam: all parameters must be fp32 on one device")
        self.t = 0
        self.counter_step = False                # True while a StepGraph / SegmentedStep runs its step function on the device step counter
        self._row_cap = {}
        self.table_rs = (os.environ.get('NR_TABLE_RS', '0') == '1') if table_rs is None else bool(table_rs)
        self.force_dist = bool(force_dist)
        self.skip_comm = os.environ.get('NR_SKIP_COMM', '0') == '1'      # measurement only: every collective skipped (ranks drift apart)
        self.comm_bytes = {}                                             # bucket name -> payload bytes per step (bench.py reports them)
        self.sched = AdamSchedule(self.lr, self.betas, dev)

        sparse_names = [n for n in self.names if any(n == s or n.endswith(s) for s in row_sparse)]
        dense = [(n, p) for n, p in named if n not in sparse_names]
        small = [(n, p) for n, p in dense if p.numel() < TABLE_MIN_NUMEL]
        tables = [(n, p) for n, p in dense if p.numel() >= TABLE_MIN_NUMEL]
        # ---- flat dense layout: [small | table 0 | table 1 ...], every region 256-byte aligned -------------------------------
        self.slices = {}
        off = 0
        for n, p in small:
            self.slices[n] = (off, off + p.numel())
            off += p.numel()
        self.regions = [_Region('small', 0, off)] if off else []
        shard_mult = _ALIGN * math.lcm(8, _world())        # a table region splits into `world` equal shards on 256-byte boundaries
        self._world0 = _world()                             # the padding above is only valid for this world size: _shard() checks it
        for n, p in tables:
            off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
            self.slices[n] = (off, off + p.numel())
            end = off + (p.numel() + shard_mult - 1) // shard_mult * shard_mult
            self.regions.append(_Region(n, off, off + p.numel(), end))
            off = end
        total = (off + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.dense_nbytes = total * 4
        self._table_region = {}
        with torch.no_grad():
            for n, p in dense:
                lo, hi = self.slices[n]
                self.flat_p[lo:hi].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[lo:hi].view(p.shape)            # the optimiser updates the parameter's own storage in place
                p.grad = self.flat_g[lo:hi].view(p.shape)
                p._nr_inplace_grad = True                            # ops.grad_target(): table scatters accumulate straight into the view
        for r in self.regions:
            if r.name != 'small':
                p = dict(dense)[r.name]
                self._table_region[id(p)] = r
                p._nr_grad_ready = self._make_ready(r)
        # ---- row-sparse tables ---------------------------------------------------------------------------------------------------------
        self.sparse = []
        for n, p in named:
            if n in sparse_names:
                if p.dim() != 2 or p.shape[1] > 1024 or not p.is_contiguous():
                    raise ValueError(f"EngineAdam: row-sparse table {n} must be a contiguous [rows, d <= 1024] matrix")
                st = _SparseTable()
                st.name, st.param, st.pending = n, p, []
                st.m, st.v = torch.zeros_like(p.data), torch.zeros_like(p.data)
                st.last = torch.zeros(p.shape[0], dtype=torch.int32, device=dev)
                st.pad_row = 0                       # nn.Embedding(padding_idx=0): row 0 receives no gradient
                p.grad = None
                p._nr_row_sink = self._make_sink(st)
                p._nr_row_sync = self._make_sync(st)
                self.sparse.append(st)
        self.param_groups = [{'lr': self.lr, 'betas': self.betas, 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
                              'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                              'params': list(range(len(self.params)))}]
        if self._module is not None and self.sparse:
            self._module.register_state_dict_pre_hook(lambda *_a, **_k: self.flush())
        if self._dist_on() and self.overlap and lib is None:
            self.enable_deferred_wgrad(True)

    # ---- hooks the engine's backward calls ------------------------------------------------------------------------------------------------
    def _make_ready(self, region):
        def ready():
            """The table's gradient is complete (its scatter is enqueued on the current stream): start the bucket's all-reduce now."""
            if self._dist_on() and self.overlap and not self.skip_comm:
                if region.work is not None:
                    raise RuntimeError(f"EngineAdam(overlap=True): the gradient of {region.name} was completed twice in one step; "
                                       "use overlap=False when a table is scattered by more than one backward call per step")
                region.work = self._start_table_exchange(region)
        return ready

    def _dist_on(self):
        return _world() > 1 or (self.force_dist and dist.is_available() and dist.is_initialized())

    def _shard(self, region):
        """(lo, hi) of this rank's shard of a table region (table_rs)."""
        w = _world()
        if w != self._world0 or (region.end - region.lo) % (w * _ALIGN):
            raise RuntimeError(f"EngineAdam: built for a process group of {self._world0} rank(s), running with {w}: the table regions were padded "
                               "for the world size at construction -- build the optimiser after init_process_group")
        sh = (region.end - region.lo) // w
        lo = region.lo + _rank() * sh
        return lo, lo + sh

    def _start_table_exchange(self, region):
        """The table bucket's gradient exchange, asynchronous on the process group's stream (RCCL's own stream on GPUs)."""
        if self.table_rs:
            lo, hi = self._shard(region)
            self.comm_bytes[region.name] = (region.end - region.lo) * 4
            return dist.reduce_scatter_tensor(self.flat_g[lo:hi], self.flat_g[region.lo:region.end], op=dist.ReduceOp.SUM, async_op=True)
        self.comm_bytes[region.name] = (region.hi - region.lo) * 4
        return dist.all_reduce(self.flat_g[region.lo:region.hi], op=dist.ReduceOp.SUM, async_op=True)

    def _make_sink(self, st):
        def sink(ids, rows):
            st.pending.append((ids.detach().reshape(-1), rows.detach().reshape(ids.numel(), -1)))
        return sink

    def _make_sync(self, st):
        def sync(ids):
            """Called by the forward before it gathers rows `ids`: replay the idle steps those rows missed.  Inside a counter-driven step
            (StepGraph / SegmentedStep capture, warm-up and eager_step set ``counter_step``) the launch is ALWAYS recorded and the kernel takes
            the step index from the device counter (its `upto <= 0` early-out covers a fresh optimiser); anywhere else -- a validation forward
            between replays -- the index is passed by value, so the rows come up to every completed step."""
            in_step = self.counter_step
            if in_step or self.t > 0:
                p = st.param.data
                self._ck(self.lib.nr_row_adam_catchup_ex(ids.data_ptr(), ids.numel(), p.data_ptr(), st.m.data_ptr(), st.v.data_ptr(),
                                                         st.last.data_ptr(), p.shape[0], p.shape[1], self.sched.table.data_ptr(), self.t,
                                                         0 if in_step else 1, self.betas[0], self.betas[1], self.eps, self._stream_fn()))
        return sync

    def _ck(self, rc):
        _capi.check(self.lib, rc)

    # ---- fault words (include/nr_engine.h: nr_set_fault_words; csrc/k_xcd.h, k_optim.h) ------------------------------------------------------
    fault_words = None

    def attach_fault_words(self):
        """Own the process's fault words (int32 [4] on the device): a persistent GRU sweep that gives up a wait sets sticky bits there and the
        optimiser kernels skip every step from then on.  Under a process group the words are max-reduced before the update kernels of every
        step, so that all ranks skip the SAME steps (a rank whose sweep failed has already poured garbage into the gradient exchange)."""
        if self.fault_words is None:
            self.fault_words = torch.zeros(4, dtype=torch.int32, device=self.device)
            self._ck(self.lib.nr_set_fault_words(self.fault_words.data_ptr()))
        return self.fault_words

    def detach_fault_words(self):
        """The library goes back to its own block (call before the optimiser is dropped: the words are this object's memory)."""
        if self.fault_words is not None:
            self._ck(self.lib.nr_set_fault_words(None))
            self.fault_words = None

    def __del__(self):
        try:
            if self.fault_words is not None:
                self.lib.nr_set_fault_words(None)
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass

    def _fault_exchange(self):
        if self.fault_words is not None and self._dist_on() and not self.skip_comm:
            return dist.all_reduce(self.fault_words, op=dist.ReduceOp.MAX, async_op=True)
        return None

    def rewind_after_fault(self):
        """SYNCHRONISES.  None when no sweep failed since the last call.  Otherwise: the index s of the first optimiser step that was skipped
        (parameters, moments and row stamps are those after step s - 1: the kernels applied nothing since); the optimiser's step count is set
        back to s - 1, pending gradients are dropped and the words cleared -- the caller repeats its steps from s on (with NR_GRU_PERSIST=0).
        Every rank of a process group gets the same answer."""
        if self.fault_words is None:
            return None
        w = self.fault_words.tolist()
        if not (w[0] or w[1]):
            return None
        s = int(w[2]) if w[2] else self.t + 1              # (no optimiser step has run since the failure: nothing to take back)
        self.discard_grads()
        self.t = s - 1
        self.fault_words.zero_()
        ops.invalidate_packed()
        return s

    # ---- the optimiser interface -------------------------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        """No-op in the steady state: ``step()`` clears the gradient buffer in the same pass that consumes it.  (Gradients of a backward
        that is NOT followed by step() are dropped by ``discard_grads()``.)"""

    def discard_grads(self):
        ops.drop_deferred()
        self.flat_g.zero_()
        for r in self.regions:
            r.work = None
        for st in self.sparse:
            st.pending.clear()

    def check_views(self):
        """autograd accumulates in place into an existing .grad, so the views must still alias the flat buffers."""
        for n, p in zip(self.names, self.params):
            if n in self.slices:
                lo, _ = self.slices[n]
                if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + lo * 4 or p.data_ptr() != self.flat_p.data_ptr() + lo * 4:
                    return False
        return True

    def _adam(self, lo, hi, scale):
        ops._timed('nr_adam_flat', lambda: self._adam_launch(lo, hi, scale))

    def _adam_launch(self, lo, hi, scale):
        self._ck(self.lib.nr_adam_flat(self.flat_p.data_ptr() + lo * 4, self.flat_g.data_ptr() + lo * 4, self.flat_m.data_ptr() + lo * 4,
                                       self.flat_v.data_ptr() + lo * 4, hi - lo, self.sched.table.data_ptr(), self.t, self.betas[0],
                                       self.betas[1], self.eps, scale, 1, self._stream_fn()))

    def enable_deferred_wgrad(self, on=True):
        """Ask the engine's backward passes to postpone their weight-gradient GEMMs past the end of loss.backward() (ops.defer_wgrad): they then
        run -- from step() / run_deferred() -- WHILE the table bucket, whose all-reduce the embedding scatter has just started, is on the wire.
        On by default under a process group with overlap; a process-wide switch (one trainer per process)."""
        ops.defer_wgrad = bool(on)
        if not on:
            ops.run_deferred()

    def step(self):
        ops.run_deferred()                       # weight-gradient phases the backward postponed (the table exchange is already in flight)
        self.t += 1
        self.sched.ensure(self.t)
        ops.invalidate_packed()                  # the kernels below rewrite parameter memory behind torch's version counters
        world = _world()
        scale = 1.0 / world
        if not self._dist_on() or self.skip_comm:
            if self.regions:
                self._adam(0, self.regions[-1].end, 1.0)                      # one launch over [small | tables]
        else:
            for r in self.regions:                                           # tables first (already in flight when overlapped), small last
                if r.name != 'small' and r.work is None:
                    r.work = self._start_table_exchange(r)
            gathered = [self._exchange_rows(st) for st in self.sparse]
            fw = self._fault_exchange()
            if fw is not None:
                fw.wait()
            for r in self.regions:
                if r.name == 'small':
                    self.comm_bytes['small'] = (r.hi - r.lo) * 4
                    r.work = dist.all_reduce(self.flat_g[r.lo:r.hi], op=dist.ReduceOp.SUM, async_op=True)
            param_gathers = []
            for r in sorted(self.regions, key=lambda r: r.name != 'small'):   # small bucket is the short message: update it while the table flies
                r.work.wait()
                r.work = None
                if self.table_rs and r.name != 'small':
                    # Adam on this rank's shard only (it also clears the shard's gradient); the other shards of the gradient buffer still
                    # hold this rank's local gradients and are cleared here; then every rank collects the updated parameters
                    lo, hi = self._shard(r)
                    self._adam(lo, hi, scale)
                    if lo > r.lo:
                        self.flat_g[r.lo:lo].zero_()
                    if hi < r.end:
                        self.flat_g[hi:r.end].zero_()
                    param_gathers.append(dist.all_gather_into_tensor(self.flat_p[r.lo:r.end], self.flat_p[lo:hi], async_op=True))
                else:
                    self._adam(r.lo, r.end, scale)
            for st, (ids, rows) in zip(self.sparse, gathered):
                self._row_step(st, ids, rows, scale)
            for w in param_gathers:                                           # the next forward reads the whole table
                w.wait()
            return
        for st in self.sparse:
            if st.pending:
                # (one (ids, rows) pair per backward is the rule: no concatenation kernels then)
                ids = st.pending[0][0] if len(st.pending) == 1 else torch.cat([i for i, _ in st.pending])
                rows = st.pending[0][1] if len(st.pending) == 1 else torch.cat([r for _, r in st.pending])
                st.pending.clear()
                self._row_step(st, ids, rows, 1.0)

    # ---- segmented form of a data-parallel step (graph.SegmentedStep): [forward + backward] | collectives | [update] -----------------------
    # The two bracketed segments are HIP graphs; the collectives between them are issued eagerly (RCCL's calls are the graph boundaries).
    # Same arithmetic as step(): all-reduce (sum) of the dense buckets, fixed-capacity all-gather of the touched rows, Adam with the
    # 1 / world scale folded in.
    def prepare_segments(self):
        """Static send / receive buffers of the touched-row exchange (their addresses are frozen into the graphs).  Needs the per-rank row
        capacity: agreed by the first eager exchange, or set_row_capacity()."""
        world = _world()
        for st in self.sparse:
            cap = self._row_cap.get(st.name)
            if cap is None:
                raise RuntimeError(f"EngineAdam.prepare_segments: no row capacity for {st.name} yet: run one eager step or set_row_capacity()")
            d = st.param.shape[1]
            st.send_ids = torch.zeros(cap, dtype=torch.int64, device=self.device)
            st.send_rows = torch.zeros(cap, d, dtype=torch.float32, device=self.device)
            st.recv_ids = torch.zeros(world * cap, dtype=torch.int64, device=self.device)
            st.recv_rows = torch.zeros(world * cap, d, dtype=torch.float32, device=self.device)

    def stage_rows(self):
        """Tail of the backward segment: this rank's (row id, gradient row) pairs into the static send buffers, padded with id 0 (the padding
        row, which the update skips) -- device work only, so that it is part of the first graph."""
        for st in self.sparse:
            cap = st.send_ids.shape[0]
            st.send_ids.zero_()
            st.send_rows.zero_()
            if st.pending:
                ids = torch.cat([i for i, _ in st.pending]).to(torch.int64)
                rows = torch.cat([r for _, r in st.pending]).to(torch.float32)
                if ids.numel() > cap:
                    raise RuntimeError(f"EngineAdam: {ids.numel()} gradient rows for {st.name}, capacity {cap}")
                st.send_ids[:ids.numel()].copy_(ids)
                st.send_rows[:ids.numel()].copy_(rows)
            st.pending.clear()

    def begin_step(self):
        """Host bookkeeping of a step whose launches live in graphs: the step index (the kernels read it from the device counter)."""
        self.t += 1
        self.sched.ensure(self.t)
        ops.invalidate_packed()

    def start_tables(self):
        """First half of the exchange: the table bucket(s) -- all-reduce, or reduce-scatter onto this rank's shard (table_rs) -- asynchronous on
        the process group's stream.  graph.SegmentedStep(overlap=True) issues it right after the segment that ends with the embedding scatter,
        so that the weight-gradient segment runs while the tables are on the wire.  Returns the work handles for exchange_all(works=...)."""
        if self.skip_comm:
            return []
        return [self._start_table_exchange(r) for r in self.regions if r.name != 'small']

    def exchange_all(self, works=None):
        """Every collective of the step (those not yet started by start_tables), issued between graph segments; returns once the current stream
        waits for all of them."""
        if self.skip_comm:
            return
        world = _world()
        if works is None:
            works = self.start_tables()
        else:
            works = list(works)
        for st in self.sparse:
            cap, d = st.send_ids.shape[0], st.param.shape[1]
            if cap:
                self.comm_bytes[st.name] = world * cap * (8 + 4 * d)
                works.append(dist.all_gather_into_tensor(st.recv_ids, st.send_ids, async_op=True))
                works.append(dist.all_gather_into_tensor(st.recv_rows, st.send_rows, async_op=True))
        for r in self.regions:
            if r.name == 'small':
                self.comm_bytes['small'] = (r.hi - r.lo) * 4
                works.append(dist.all_reduce(self.flat_g[r.lo:r.hi], op=dist.ReduceOp.SUM, async_op=True))
        fw = self._fault_exchange()
        if fw is not None:
            works.append(fw)
        for w in works:
            w.wait()

    def apply_all(self):
        """The update launches of the step (last graph segment): fused Adam over every dense bucket with the 1 / world scale -- a table bucket
        exchanged by reduce-scatter (table_rs) on this rank's shard only, the other shards' (local) gradients cleared --, the row-sparse step
        over the gathered rows.  With table_rs the caller then collects the updated table: gather_tables() (a collective, outside the graph)."""
        scale = 1.0 / _world()
        for r in self.regions:
            if self.table_rs and r.name != 'small':
                lo, hi = self._shard(r)
                self._adam(lo, hi, scale)
                if lo > r.lo:
                    self.flat_g[r.lo:lo].zero_()
                if hi < r.end:
                    self.flat_g[hi:r.end].zero_()
            else:
                self._adam(r.lo, r.end, scale)
        for st in self.sparse:
            if st.recv_ids.numel():
                self._row_step(st, st.recv_ids, st.recv_rows, scale)

    def gather_tables(self):
        """table_rs: every rank collects the table parameters the other ranks' shards of the Adam pass updated (all-gather; the same bytes as the
        second half of an all-reduce).  A no-op for the all-reduce form."""
        if not self.table_rs or self.skip_comm:
            return
        works = []
        for r in self.regions:
            if r.name != 'small':
                lo, hi = self._shard(r)
                works.append(dist.all_gather_into_tensor(self.flat_p[r.lo:r.end], self.flat_p[lo:hi], async_op=True))
        for w in works:
            w.wait()

    def _exchange_rows(self, st):
        """All ranks' (row id, gradient row) pairs, in rank order: B x (8 + 4 d) bytes per rank instead of a table-sized all-reduce."""
        world = _world()
        d = st.param.shape[1]
        if st.pending:
            ids = torch.cat([i for i, _ in st.pending]).to(torch.int64)
            rows = torch.cat([r for _, r in st.pending]).to(torch.float32).contiguous()
        else:
            ids = torch.zeros(0, dtype=torch.int64, device=self.device)
            rows = torch.zeros(0, d, dtype=torch.float32, device=self.device)
        st.pending.clear()
        # The protocol must not depend on rank-local state (a rank that picked another collective than its peers would hang or corrupt the
        # exchange): every rank, every step, all-gathers exactly `cap` (id, row) pairs, padded with id 0 -- the padding row of
        # nn.Embedding(padding_idx=0), which the update skips.  `cap` is agreed once, at the first exchange of the table (one all-gather of
        # the local counts: the only host round trip), as the largest per-rank count -- the fixed per-GPU batch of
        # DataLoader(drop_last=True) (src/train.py:118-124); a shorter batch is padded, a longer one is an error on the rank that sees it.
        n_local = ids.numel()
        cap = self._row_cap.get(st.name)
        if cap is None:
            n = torch.tensor([n_local], dtype=torch.int64, device=self.device)
            counts = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(counts, n)
            cap = self._row_cap[st.name] = max(int(c.item()) for c in counts)
        if n_local > cap:
            raise RuntimeError(f"EngineAdam: {n_local} gradient rows for {st.name} on rank {_rank()}, but the ranks agreed on at most {cap} per "
                               "step at the first exchange; keep the per-rank batch fixed or call set_row_capacity() on every rank")
        if cap == 0:
            return ids, rows
        if n_local < cap:
            pid = torch.zeros(cap, dtype=torch.int64, device=self.device)
            prow = torch.zeros(cap, d, dtype=torch.float32, device=self.device)
            pid[:n_local] = ids
            prow[:n_local] = rows
            ids, rows = pid, prow
        self.comm_bytes[st.name] = world * cap * (8 + 4 * d)
        gid = torch.empty(world * cap, dtype=torch.int64, device=self.device)
        grow = torch.empty(world * cap, d, dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(gid, ids.contiguous())
        dist.all_gather_into_tensor(grow, rows)
        return gid, grow

    def set_row_capacity(self, cap, name=None):
        """Fix the per-rank, per-step capacity of the touched-row exchange (call with the same value on every rank, e.g. the largest batch)."""
        for st in self.sparse:
            if name is None or st.name == name:
                self._row_cap[st.name] = int(cap)

    def _row_step(self, st, ids, rows, scale):
        if ids.numel() == 0:
            return
        ids = ids.to(torch.int64).contiguous()
        n, p = ids.numel(), st.param.data
        ws_bytes = self.lib.nr_sort_ids_workspace(n, p.shape[0])
        ids_sorted, perm = torch.empty_like(ids), torch.empty_like(ids)
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=ids.device)
        self._ck(self.lib.nr_sort_ids(ids.data_ptr(), n, p.shape[0], ids_sorted.data_ptr(), perm.data_ptr(), ws.data_ptr(), ws_bytes,
                                      self._stream_fn()))
        rows = rows.contiguous()
        self._ck(self.lib.nr_row_adam_step(ids_sorted.data_ptr(), perm.data_ptr(), n, rows.data_ptr(), rows.shape[1], p.data_ptr(),
                                           st.m.data_ptr(), st.v.data_ptr(), st.last.data_ptr(), p.shape[0], p.shape[1],
                                           self.sched.table.data_ptr(), self.t, self.betas[0], self.betas[1], self.eps, scale, st.pad_row,
                                           self._stream_fn()))

    def flush(self):
        """Bring every row of the row-sparse tables up to date with the steps taken so far (before the table is read as a whole)."""
        if self.t == 0:
            return
        for st in self.sparse:
            p = st.param.data
            self._ck(self.lib.nr_row_adam_flush(p.data_ptr(), st.m.data_ptr(), st.v.data_ptr(), st.last.data_ptr(), p.shape[0], p.shape[1],
                                                self.sched.table.data_ptr(), self.t, self.betas[0], self.betas[1], self.eps, self._stream_fn()))

    # ---- torch.optim.Adam-compatible state (src/train.py:151-152,268-275) ---------------------------------------------------------
    def _moments(self, i):
        n, p = self.names[i], self.params[i]
        if n in self.slices:
            lo, hi = self.slices[n]
            return self.flat_m[lo:hi].view(p.shape), self.flat_v[lo:hi].view(p.shape)
        st = next(s for s in self.sparse if s.name == n)
        return st.m, st.v

    def state_is_sharded(self):
        """True when the Adam moments of the table regions live sharded across the ranks (table_rs under a process group): reading them as a
        whole is then a COLLECTIVE -- every rank must call gather_state() before any rank calls state_dict()."""
        return bool(self.table_rs and self._dist_on() and not self.skip_comm)

    def gather_state(self):
        """COLLECTIVE (all ranks, same order): collect the other ranks' shards of the table moments, so that a following state_dict() on
        any subset of the ranks (rank 0 writing a checkpoint, train_fast.py) is purely local.  A no-op when nothing is sharded."""
        if not self.state_is_sharded():
            return
        for r in self.regions:
            if r.name != 'small':
                lo, hi = self._shard(r)
                for buf in (self.flat_m, self.flat_v):
                    dist.all_gather_into_tensor(buf[r.lo:r.end], buf[lo:hi].clone())
        self._gathered_t = self.t

    def state_dict(self):
        """torch.optim.Adam's state layout.  Never issues a collective itself: with sharded moments (state_is_sharded) the caller must have
        run gather_state() on EVERY rank since the last step -- a rank-0-only state_dict() that started all-gathers on its own would pair
        them with whatever collective the other ranks issue next (hang / corrupted exchange)."""
        self.flush()
        if self.state_is_sharded() and getattr(self, '_gathered_t', -1) != self.t:
            raise RuntimeError("EngineAdam.state_dict(): the table moments are sharded across ranks (table_rs); call gather_state() on "
                               "every rank first (it is a collective), then state_dict() where the checkpoint is written")
        state = {}
        if self.t > 0:
            for i in range(len(self.params)):
                m, v = self._moments(i)
                state[i] = {'step': torch.tensor(float(self.t)), 'exp_avg': m.detach().clone(), 'exp_avg_sq': v.detach().clone()}
        return {'state': state, 'param_groups': [dict(g) for g in self.param_groups]}

    def load_state_dict(self, sd):
        groups = sd['param_groups']
        n = sum(len(g['params']) for g in groups)
        if n != len(self.params):
            raise ValueError(f"EngineAdam.load_state_dict: {n} parameters in the checkpoint, {len(self.params)} in the model")
        g0 = groups[0]
        if (float(g0['lr']), tuple(map(float, g0['betas'])), float(g0['eps'])) != (self.lr, self.betas, self.eps):
            self.lr, self.betas, self.eps = float(g0['lr']), tuple(map(float, g0['betas'])), float(g0['eps'])
            self.sched = AdamSchedule(self.lr, self.betas, self.device)
            self.param_groups[0].update(lr=self.lr, betas=self.betas, eps=self.eps)
        steps = {int(float(s['step'])) for s in sd['state'].values()}
        if len(steps) > 1:
            raise ValueError("EngineAdam.load_state_dict: parameters with different step counts are not supported")
        self.t = steps.pop() if steps else 0
        self.sched.ensure(self.t)
        with torch.no_grad():
            for i in range(len(self.params)):
                m, v = self._moments(i)
                s = sd['state'].get(i)
                if s is None:
                    m.zero_()
                    v.zero_()
                else:
                    m.copy_(s['exp_avg'].to(self.device))
                    v.copy_(s['exp_avg_sq'].to(self.device))
            for st in self.sparse:                      # rows with optimiser state are current as of the checkpoint's step
                has = ((st.m != 0) | (st.v != 0)).any(dim=1)
                st.last.copy_(has.to(torch.int32) * self.t)
