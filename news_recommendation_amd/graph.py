"""One HIP graph for the whole training step (forward + backward + fused Adam) -- the loop body of src/train.py:183-233.

At B = 512 the NRMS step is ~60 kernel launches of 5 - 900 us each; issued one by one they leave ~0.3 ms of idle gaps per step on the
GPU and cost 1.2 - 1.4 ms of host time per step.  Captured once and replayed, the step costs one launch.

What has to be true for a captured step to remain a TRAINING step when replayed:
  * new dropout masks every step      -> the kernels fold a device-resident step counter into their keys (nr_set_step_counter,
                                         csrc/nr_common.h drop_resolve); the seeds frozen into the graph are only a base;
  * Adam's bias-correction step index -> nr_adam_flat reads it from the same counter (csrc/k_optim.h);
  * inputs                            -> copied into static device buffers before each replay;
  * no host round trips, no allocation outside the graph's pool -- true of the engine's step by construction (ids resident in HBM,
    gradients accumulate into the optimiser's flat buffers, the token-id sort runs on a forked side stream that joins before the scatter).
The counter is bumped by the graph's first node, so replay k uses counter value t0 + k: exactly what the eager loop below (`eager_step`)
does with one nr_step_counter_add launch per step -- which is how the tests hold replays to the eager path bit for bit.

Scope: single process (world = 1); NRMS, NAML and LSTUR.  LSTUR's step is capturable when its inputs are device tensors: the history
lengths stay on the device (the recurrence then always runs all N steps; finished samples keep their state, same result), the whole-row
user mask is drawn by the engine's counter-based generator (nr_dropout_mask, site 3) instead of the host's, and the row-sparse Adam of the
user table reads its step index from the same counter (nr_row_adam_catchup / nr_row_adam_step, csrc/k_optim.h); its (row id, gradient row)
list never leaves the device.  Data-parallel steps are captured in SEGMENTS around their RCCL calls (SegmentedStep below).
"""
import torch

from . import _capi, ops


_attached = []        # counter tensors the library currently points at (at most one: the counter is process-global)


class _CounterStep:
    """Marks the optimiser as inside a counter-driven step for the duration of a step function (optim.EngineAdam._make_sync: the row-sparse
    catch-up then takes its step index from the device counter; outside, by value)."""

    def __init__(self, opt):
        self.opt = opt

    def __enter__(self):
        self.prev = getattr(self.opt, 'counter_step', False)
        self.opt.counter_step = True

    def __exit__(self, *exc):
        self.opt.counter_step = self.prev


class StepGraph:
    """``g = StepGraph(step_fn, example_inputs, optimizer)``; ``loss = g(*inputs)`` runs one training step.

    step_fn(*inputs) must run forward, ``loss.backward()`` and ``optimizer.step()`` and return the (device) loss tensor; `optimizer` is
    an ``optim.EngineAdam`` (row-sparse tables included: their kernels follow the device step counter).  ``max_steps`` sizes the optimiser's per-step scalar table once (its address is
    frozen into the graph)."""

    def __init__(self, step_fn, example_inputs, optimizer, warmup=2, max_steps=1 << 20):
        if _attached:
            raise RuntimeError("StepGraph: another StepGraph of this process still has its step counter attached (close() it first): the "
                               "counter is process-global, every dropout kernel and nr_adam_flat reads it")
        if optimizer._dist_on():
            raise NotImplementedError("StepGraph: data-parallel steps are not captured (the collectives are graph boundaries)")
        self.lib = _capi.load()
        self.opt = optimizer
        self.step_fn = step_fn
        dev = optimizer.device
        self.static = [torch.empty_like(x) for x in example_inputs]
        for s, x in zip(self.static, example_inputs):
            s.copy_(x)
        optimizer.sched.ensure(optimizer.t + max_steps)
        self.max_t = optimizer.t + max_steps
        # the device counter mirrors optimizer.t: both are bumped at the START of a step
        self.ctr = torch.full((1,), optimizer.t, dtype=torch.int32, device=dev)
        _capi.check(self.lib, self.lib.nr_set_step_counter(self.ctr.data_ptr()))
        _attached.append(self.ctr)              # module-level holder: the library keeps a RAW pointer to this tensor until close()
        self._closed = False
        # warm-up on a side stream (lazy initialisation, LDS opt-ins, packed-operand caches, workspace growth) -- these are real steps
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.eager_step(*self.static)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        ops.invalidate_packed()                 # the captured step must contain its own packing launches
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            _capi.check(self.lib, self.lib.nr_step_counter_add(self.ctr.data_ptr(), 1, torch.cuda.current_stream().cuda_stream))
            with _CounterStep(self.opt):
                self.loss = step_fn(*self.static)
        # the capture ran the Python side of one step (optimizer.t advanced) without executing a kernel: take it back
        self.opt.t -= 1
        ops.invalidate_packed()                 # (operands cached during the capture were recorded, not computed)

    def eager_step(self, *inputs):
        """The same step without the graph, on the same counter protocol (used for warm-up and as the reference in tests)."""
        _capi.check(self.lib, self.lib.nr_step_counter_add(self.ctr.data_ptr(), 1, torch.cuda.current_stream().cuda_stream))
        with _CounterStep(self.opt):
            return self.step_fn(*inputs)

    def __call__(self, *inputs):
        if self.opt.t + 1 > self.max_t:
            raise RuntimeError("StepGraph: max_steps exhausted (the optimiser's step table is sized at capture)")
        for s, x in zip(self.static, inputs):
            if s.data_ptr() != x.data_ptr():
                s.copy_(x, non_blocking=True)
        self.graph.replay()
        self.opt.t += 1
        ops.invalidate_packed()                 # parameters changed behind torch's version counters (as after an eager optimizer.step())
        return self.loss

    def close(self):
        """Detach the counter: later launches of this process take seeds / step indices by value again.  Idempotent; also run by the
        context-manager exit and the finaliser, so that a StepGraph dropped after an exception never leaves the library pointing at freed
        memory (the holder keeps the counter tensor alive until then)."""
        if getattr(self, '_closed', True):
            return
        # closed -- and the counter tensor released -- only once the library no longer points at it: if the synchronisation or the detach
        # raises, the holder keeps the tensor alive and a later close() can retry
        torch.cuda.synchronize(self.opt.device)             # no replay in flight still reads the counter
        _capi.check(self.lib, self.lib.nr_set_step_counter(None))
        self._closed = True
        _attached[:] = [c for c in _attached if c is not self.ctr]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SegmentedStep:
    """A DATA-PARALLEL training step as HIP graphs with the RCCL collectives between them.  Two issue modes:

    overlap=True (default)  THREE segments, the table exchange in flight under the second one:
        graph A   counter bump, forward, backward up to and including the embedding scatter (the backward passes postpone their weight-gradient
                  GEMMs: ops.defer_wgrad)
        eager     table bucket(s): all-reduce -- or reduce-scatter onto this rank's shard (table_rs) -- started on RCCL's stream
        graph B   the postponed weight-gradient GEMMs (ops.run_deferred), touched rows staged for the exchange       <- runs while the tables fly
        eager     all-gather of the touched rows, all-reduce of the small bucket (+ the fault words), wait for everything
        graph C   fused Adam over the dense buckets (table_rs: this rank's shard), row-sparse Adam over the gathered rows
        eager     table_rs only: all-gather of the updated table
    overlap=False           round 4's TWO segments: [forward + whole backward + staging] | every collective | [Adam]: the table all-reduce
                  (85 - 156 MB) is then fully exposed (kept for A/B: bench.py --gpus N reports exposed_comm_ms for both).

    Issued kernel by kernel, the step costs ~2 ms of host time; in either form the host issues two or three graph launches and three to six
    collectives.  Same arithmetic as ``EngineAdam.step()``; ``eager_step`` is that path on the same counter protocol (the tests hold the two
    together: tests/test_rccl_gpu.py over RCCL, tests/test_dist_cpu.py for the exchange protocol over gloo)."""

    def __init__(self, fwd_bwd_fn, example_inputs, optimizer, warmup=2, max_steps=1 << 20, overlap=True):
        if not optimizer._dist_on():
            raise RuntimeError("SegmentedStep: no process group (single process: use StepGraph)")
        if _attached:
            raise RuntimeError("SegmentedStep: another captured step of this process still has its step counter attached (close() it first)")
        self.lib = _capi.load()
        self.opt = optimizer
        self.fn = fwd_bwd_fn
        self.overlap = bool(overlap)
        dev = optimizer.device
        optimizer.overlap = False            # no collective is issued from INSIDE the backward: they sit between the segments
        optimizer.enable_deferred_wgrad(self.overlap)
        self.static = [torch.empty_like(x) for x in example_inputs]
        for s, x in zip(self.static, example_inputs):
            s.copy_(x)
        optimizer.sched.ensure(optimizer.t + max_steps)
        self.max_t = optimizer.t + max_steps
        self.ctr = torch.full((1,), optimizer.t, dtype=torch.int32, device=dev)
        _capi.check(self.lib, self.lib.nr_set_step_counter(self.ctr.data_ptr()))
        _attached.append(self.ctr)
        self._closed = False
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # real steps; the first one also agrees the per-rank row capacity of the touched-row exchange
                self.eager_step(*self.static)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        optimizer.prepare_segments()
        ops.invalidate_packed()
        ops.drop_deferred()
        # thread_local: RCCL's watchdog thread polls events of its own while this thread captures
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):
            _capi.check(self.lib, self.lib.nr_step_counter_add(self.ctr.data_ptr(), 1, torch.cuda.current_stream().cuda_stream))
            with _CounterStep(optimizer):
                self.loss = fwd_bwd_fn(*self.static)
            if not self.overlap:
                optimizer.stage_rows()
        self.graph_w = None
        if self.overlap:
            # the weight-gradient phases the backward queued (closures over tensors of graph A's pool, which the two graphs share)
            self.graph_w = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_w, pool=self.graph_a.pool(), capture_error_mode="thread_local"):
                ops.run_deferred()
                optimizer.stage_rows()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode="thread_local"):
            optimizer.apply_all()
        # the capture RECORDED the packing launches without running them: the operands it left in the cache hold no data yet (a replay fills
        # them; an eager step taken now must not find them)
        ops.invalidate_packed()

    def eager_step(self, *inputs):
        _capi.check(self.lib, self.lib.nr_step_counter_add(self.ctr.data_ptr(), 1, torch.cuda.current_stream().cuda_stream))
        with _CounterStep(self.opt):
            loss = self.fn(*inputs)
        self.opt.step()                      # (runs the postponed weight-gradient phases first)
        return loss

    def __call__(self, *inputs):
        if self.opt.t + 1 > self.max_t:
            raise RuntimeError("SegmentedStep: max_steps exhausted (the optimiser's step table is sized at capture)")
        for s, x in zip(self.static, inputs):
            if s.data_ptr() != x.data_ptr():
                s.copy_(x, non_blocking=True)
        self.graph_a.replay()
        self.opt.begin_step()
        if self.overlap:
            works = self.opt.start_tables()  # on RCCL's stream, behind graph A
            self.graph_w.replay()            # weight-gradient GEMMs + row staging under the table exchange
            self.opt.exchange_all(works)
        else:
            self.opt.exchange_all()
        self.graph_b.replay()
        self.opt.gather_tables()
        ops.invalidate_packed()
        return self.loss

    def close(self):
        StepGraph.close(self)
        self.opt.enable_deferred_wgrad(False)

    __enter__ = StepGraph.__enter__
    __exit__ = StepGraph.__exit__

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
