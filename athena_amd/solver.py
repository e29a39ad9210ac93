"""Training / evaluation loops (mirror of the reference's athena/solver.py:45-218).

``BaseSolver.train_step`` = forward, loss, backward, per-tensor clip + Adam (one fused kernel).
``HorovodSolver`` is the data-parallel solver: one process per GPU under ``torchrun``, the flat
gradient buffer is averaged across ranks with bucketed all-reduces on RCCL
(``torch.distributed`` backend "nccl") in place of ``hvd.DistributedGradientTape``.
"""
import contextlib
import gc
import logging
import os

import torch
import torch.distributed as dist

from . import ops
from .utils.hparam import register_and_parse_hparams
from .utils.metric_check import MetricChecker


class BaseSolver:
    """Base Solver."""
    default_config = {
        "clip_norm": 100.0,
        "log_interval": 10,
        "enable_tf_function": True
    }

    def __init__(self, model, optimizer, sample_signature=None, eval_sample_signature=None, config=None, **kwargs):
        self.model = model
        self.optimizer = optimizer
        if getattr(optimizer, "store", None) is None:
            optimizer.attach(model.store)
        self.metric_checker = MetricChecker(self.optimizer)
        self.sample_signature = sample_signature
        self.eval_sample_signature = eval_sample_signature
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        self.world_size = 1
        self.rank = 0

    @staticmethod
    def initialize_devices(solver_gpus=None):
        """Select the GPU of this process (solver.py:63-76)."""
        if torch.cuda.is_available():
            idx = solver_gpus[0] if solver_gpus else 0
            torch.cuda.set_device(idx)

    @staticmethod
    def clip_by_norm(grads, norm):
        """tf.clip_by_norm per tensor on a list of gradients (solver.py:78-87).  The training step
        itself uses the fused clip inside the Adam kernel; this stays for API compatibility."""
        if norm <= 0:
            return grads
        out = []
        for g in grads:
            if g is None:
                out.append(None)
                continue
            n = torch.linalg.vector_norm(g.float())
            out.append(g * (norm / torch.clamp(n, min=norm)).to(g.dtype))
        return out

    # ------------------------------------------------------------------ one step
    def _forward_backward(self, samples):
        # nothing here looks at the gradient of the logits, so the losses may hand it to the
        # vocabulary projections in bf16
        with ops.bf16_grad_handover():
            outputs = self.model(samples, training=True)
            loss, metrics = self.model.get_loss(outputs, samples, training=True)
            total_loss = sum(list(loss.values())) if isinstance(loss, dict) else loss
            # (the root gradient is a constant one that already sits on the device: no fill per step)
            one = ops.const_scalar(total_loss.device, 1.0) if total_loss.is_cuda and total_loss.dim() == 0 and \
                total_loss.dtype == torch.float32 else None
            total_loss.backward(one)
        return loss, metrics

    def _reduce_gradients(self):
        """Hook for data parallelism; returns the factor the optimizer applies to the gradients."""
        return 1.0

    def _flush_metrics(self):
        for m in (getattr(self.model, "metric", None), getattr(getattr(self.model, "model", None), "metric", None)):
            if m is not None and hasattr(m, "flush"):
                m.flush()

    def _metric_objects(self):
        out = []
        for m in (getattr(self.model, "metric", None), getattr(getattr(self.model, "model", None), "metric", None)):
            if m is not None and hasattr(m, "_pending") and m not in out:
                out.append(m)
        return out

    def _update(self, scale):
        self.optimizer.apply_gradients(clip_norm=self.hparams.clip_norm, grad_scale=scale)
        ops.advance_rng(self.model.store.rng_state)

    def _eager_step(self, samples):
        loss, metrics = self._forward_backward(samples)
        scale = self._reduce_gradients()
        self._update(scale)
        self._flush_metrics()
        return loss, metrics

    def train_step(self, samples):
        """train the model 1 step (solver.py:89-99).

        ``enable_tf_function`` (the reference compiles the step with tf.function, solver.py:91-93)
        selects HIP-graph replay here: the first step on a new batch shape runs eagerly, the second
        one records forward + backward + update into a HIP graph, and every later step copies the
        batch into the recorded input buffers and replays the graph.  The returned loss tensors are
        the graph's output buffers: read them before the next step."""
        if not self._graphable(samples):
            return self._eager_step(samples)
        key = _shape_key(samples)
        state = self._graphs.pop(key, None)
        if state is None:                    # first sight of this shape: plain step (also the warm-up)
            self._graphs[key] = False
            self._trim_graphs()
            return self._eager_step(samples)
        if state is False:
            state = self._record(samples)
        self._graphs[key] = state            # most recently used last
        self._trim_graphs()
        return state.run(samples)

    # Recorded steps kept at a time.  Every recorded shape holds the intermediates of a whole step in
    # the graphs' memory pool (shared by all graphs of this solver, so the blocks of an evicted
    # graph serve the next recording); batches padded to their own longest utterance produce
    # hundreds of shapes per epoch, of which only the most recently used ones stay recorded.
    MAX_GRAPHS = int(os.environ.get("ATHENA_MAX_GRAPHS", "24"))

    def _trim_graphs(self):
        while len(self._graphs) > max(self.MAX_GRAPHS, 1):
            oldest = next(iter(self._graphs))
            del self._graphs[oldest]         # a shape seen again later starts over with an eager step

    def _graph_pool(self):
        if getattr(self, "_pool", None) is None:
            self._pool = torch.cuda.graph_pool_handle()
        return self._pool

    def _record(self, samples):
        return _GraphStep(self, samples)

    def _graphable(self, samples):
        if not getattr(self.hparams, "enable_tf_function", False) or not isinstance(samples, dict):
            return False
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        return all(isinstance(v, torch.Tensor) and v.is_cuda for v in samples.values())

    def train(self, dataset, total_batches=-1):
        """Update the model in 1 epoch (solver.py:101-113)."""
        for batch, samples in enumerate(dataset):
            if 0 <= total_batches <= batch:
                break
            samples = self.model.prepare_samples(samples)
            loss, metrics = self.train_step(samples)
            if batch % self.hparams.log_interval == 0:
                # every rank drops its queued metric updates here; only rank 0 reports them
                if self.rank == 0:
                    logging.info(self.metric_checker(loss, metrics))
                self.model.reset_metrics()

    def evaluate_step(self, samples):
        """evaluate the model 1 step (solver.py:115-120)."""
        with torch.no_grad():
            outputs = self.model(samples, training=False)
            loss, metrics = self.model.get_loss(outputs, samples, training=False)
        self._flush_metrics()
        return loss, metrics

    def evaluate(self, dataset, epoch=0):
        """evaluate the model (solver.py:122-140)."""
        loss_sum, n = 0.0, 0
        loss, metrics = None, None
        self.model.reset_metrics()
        for batch, samples in enumerate(dataset):
            samples = self.model.prepare_samples(samples)
            loss, metrics = self.evaluate_step(samples)
            if batch % self.hparams.log_interval == 0 and self.rank == 0:
                logging.info(self.metric_checker(loss, metrics, -2))
            total_loss = sum(list(loss.values())) if isinstance(loss, dict) else loss
            loss_sum += float(total_loss)
            n += 1
        mean_loss = loss_sum / max(n, 1)
        if metrics is not None:
            metrics = {k: float(v) for k, v in metrics.items()}
        if self.rank == 0:
            logging.info(self.metric_checker(mean_loss, metrics, evaluate_epoch=epoch))
        self.model.reset_metrics()
        return mean_loss, metrics


# other host threads (the RCCL watchdog polling its events) keep making HIP calls while a step is
# being recorded: only calls of the recording thread itself may invalidate the capture
_CAPTURE_MODE = "thread_local"


@contextlib.contextmanager
def _capture(graph, pool):
    """Record into ``graph``.  The garbage collector stays off while the stream is capturing: a
    collection in the middle of a recording may run the destructor of an evicted step's graphs (they
    sit in reference cycles with their solver), and releasing a graph and its memory blocks from
    inside another graph's capture aborts the process."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, pool=pool, capture_error_mode=_CAPTURE_MODE):
            yield
    finally:
        if was_enabled:
            gc.enable()


def _copy_inputs(buffers, samples):
    """The batch into the buffers a recorded step reads: device tensors as one launch, anything else
    (a host batch) by the runtime's copies."""
    pairs = []
    for k, v in buffers.items():
        src = samples[k]
        if src.is_cuda and v.is_cuda and src.dtype == v.dtype and src.shape == v.shape and src.is_contiguous():
            if src.data_ptr() != v.data_ptr():
                pairs.append((src, v))
        else:
            v.copy_(src, non_blocking=True)
    if pairs:
        ops.copy_many(pairs)


def _shape_key(samples):
    return tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(samples.items()))


class _GraphStep:
    """One training step recorded as HIP graphs for one batch shape.

    Single process: one graph holds forward, backward, clip + Adam and the dropout-counter bump.
    Data parallel: the gradient all-reduce stays outside (between a forward/backward graph and an
    update graph), so no collective is recorded.  Metric hand-over: the arg-max kernels recorded in
    the graph leave their ids in fixed device buffers; after every replay these are copied to fresh
    pinned host buffers and queued for the usual deferred host-side scoring."""

    def __init__(self, solver, samples):
        self.solver = solver
        self.inputs = {k: v.clone() for k, v in samples.items()}
        metrics_objs = solver._metric_objects()
        before = [len(m._pending) for m in metrics_objs]
        split = solver.world_size > 1 or getattr(solver, "_force_exchange", False)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # the recorded forward pass always starts by rebuilding the bf16 weight copies: whether the
        # host flag happens to be set while recording must not decide what every replay does
        solver.model.store.shadows_dirty = True
        self.update_graph = None
        with ops.step_tables(self, next(iter(self.inputs.values())).device):
            with _capture(self.graph, solver._graph_pool()):
                self.loss, self.metrics = solver._forward_backward(self.inputs)
                if not split:
                    solver._update(1.0)
            if split:
                self.update_graph = torch.cuda.CUDAGraph()
                with _capture(self.update_graph, solver._graph_pool()):
                    solver._update(1.0 / solver.world_size)
        # the deferred scorers queued while recording become templates
        self.templates = []
        for m, n in zip(metrics_objs, before):
            self.templates.append((m, m._pending[n:]))
            del m._pending[n:]

    def run(self, samples):
        from .metrics import _Deferred
        _copy_inputs(self.inputs, samples)
        self.graph.replay()
        if self.update_graph is not None:
            self.solver._reduce_gradients()
            self.update_graph.replay()
        # the replayed update changed the weights behind the host's back
        self.solver.model.store.shadows_dirty = True
        for m, items in self.templates:
            for t in items:
                m._pending.append(_Deferred(t.owner, list(t.tensors), t.score).to_host())
        return self.loss, self.metrics


class _PipelinedGraphStep:
    """The data-parallel step as a chain of HIP graphs with the gradient exchange in between.

    The backward pass is cut at a few activations (decoder | encoder thirds | front end).  Each piece
    is its own graph; when a piece has been replayed, the gradients of the parameters it covers are
    final -- they form one contiguous range of the flat gradient buffer -- and their all-reduce is
    started on RCCL's stream while the next piece is replayed.  Only the exchange of the last range
    (front end + first encoder layers) is not hidden behind compute.  The update graph (clip + Adam
    with 1/world_size) runs after all exchanges."""

    def __init__(self, solver, samples):
        self.solver = solver
        store = solver.model.store
        self.inputs = {k: v.clone() for k, v in samples.items()}
        metrics_objs = solver._metric_objects()
        before = [len(m._pending) for m in metrics_objs]
        torch.cuda.synchronize()
        self.graphs = []
        with ops.step_tables(self, store.grads.device):
            self._record_chain(solver, store)
        self.templates = []
        for m, n in zip(metrics_objs, before):
            self.templates.append((m, m._pending[n:]))
            del m._pending[n:]

    def _record_chain(self, solver, store):
        g0 = torch.cuda.CUDAGraph()
        store.collect_cuts = True
        store.shadows_dirty = True          # the refresh is always part of the recorded forward pass
        try:
            with _capture(g0, solver._graph_pool()), \
                    ops.bf16_grad_handover():
                outputs = solver.model(self.inputs, training=True)
                self.loss, self.metrics = solver.model.get_loss(outputs, self.inputs, training=True)
                total = sum(list(self.loss.values())) if isinstance(self.loss, dict) else self.loss
                cuts = dict(store.cuts)
                order = [t for t, _ in store.cuts]
                enc_cuts = [t for t in order if t.startswith("encoder.")]          # forward order
                g_mem, g_y = torch.autograd.grad(total, [cuts["memory"], cuts["y_net"]], retain_graph=True)
        finally:
            store.collect_cuts = False
            store.cuts = []
        self.graphs.append(g0)
        self.ranges = _piece_ranges(store, [int(t.split(".")[1]) for t in enc_cuts])
        # encoder pieces, from the top of the stack down
        up, g_up = cuts["memory"], g_mem
        for tag in reversed(enc_cuts):
            g = torch.cuda.CUDAGraph()
            with _capture(g, g0.pool()):
                (g_dn,) = torch.autograd.grad([up], [cuts[tag]], [g_up], retain_graph=True)
            self.graphs.append(g)
            up, g_up = cuts[tag], g_dn
        g = torch.cuda.CUDAGraph()
        with _capture(g, g0.pool()):
            torch.autograd.backward([up, cuts["y_net"]], [g_up, g_y])
        self.graphs.append(g)
        assert len(self.graphs) == len(self.ranges)
        self.update_graph = torch.cuda.CUDAGraph()
        with _capture(self.update_graph, g0.pool()):
            solver._update(1.0 / max(solver.world_size, 1))
        self._keep = (cuts, g_mem, g_y, up, g_up)          # the graphs' buffers must outlive them

    def run(self, samples):
        from .metrics import _Deferred
        _copy_inputs(self.inputs, samples)
        grads = self.solver.model.store.grads
        step = max(int(self.solver.BUCKET_MB * (1 << 20) / 4), 1)
        works = []
        for g, (a, b) in zip(self.graphs, self.ranges):
            g.replay()
            for lo in range(a, b, step):
                works.append(self.solver._exchange_async(grads[lo:min(lo + step, b)]))
        for w in works:
            if w is not None:
                w.wait()
        self.update_graph.replay()
        self.solver.model.store.shadows_dirty = True
        for m, items in self.templates:
            for t in items:
                m._pending.append(_Deferred(t.owner, list(t.tensors), t.score).to_host())
        return self.loss, self.metrics


class _OneGraphPipelinedStep:
    """The data-parallel step as ONE HIP graph with the gradient exchange inside it.

    Same cuts of the backward pass as _PipelinedGraphStep (decoder | encoder thirds | front end), but
    the all-reduce of each finished range is recorded where it is issued, on a side stream: RCCL's
    kernels become nodes of the graph, a branch beside the pieces of the backward pass that follow,
    joined in front of the update.  One graph launch per step instead of six, so the step costs what
    the single-process step costs plus whatever of the last exchange is not hidden, however slow the
    host is.  The collectives go through the solver's own communicator (utils/rccl.py): one issued
    through torch.distributed cannot be recorded safely, its watchdog thread polls events that were
    recorded while capturing."""

    def __init__(self, solver, samples):
        self.solver = solver
        store = solver.model.store
        self.inputs = {k: v.clone() for k, v in samples.items()}
        metrics_objs = solver._metric_objects()
        before = [len(m._pending) for m in metrics_objs]
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        grads = store.grads
        step = max(int(solver.BUCKET_MB * (1 << 20) / 4), 1)
        works = []

        def exchange(a, b):
            for lo in range(a, b, step):
                works.append(solver._exchange_async(grads[lo:min(lo + step, b)]))

        from .layers.fused import _WgradQueue
        # (ATHENA_DP_SIDE_TABLES=1: each piece's weight-gradient table on the exchange stream instead of
        # the launch stream.  Not the default: a table whose contractions are split fills the chip by
        # itself, and beside the backward pass it only takes compute units away from kernels that need
        # all of them at once.)
        side_tables = solver._rccl is not None and os.environ.get("ATHENA_DP_SIDE_TABLES", "0") == "1"
        if side_tables and solver._side is None:
            solver._side = torch.cuda.Stream()
        store.collect_cuts = True
        store.shadows_dirty = True          # the refresh is always part of the recorded forward pass
        try:
            # each piece's weight-gradient table runs on the side stream, in front of the piece's exchange
            _WgradQueue.side = solver._side if side_tables else None
            with ops.step_tables(self, grads.device), _capture(self.graph, solver._graph_pool()), \
                    ops.bf16_grad_handover():
                outputs = solver.model(self.inputs, training=True)
                self.loss, self.metrics = solver.model.get_loss(outputs, self.inputs, training=True)
                total = sum(list(self.loss.values())) if isinstance(self.loss, dict) else self.loss
                cuts = dict(store.cuts)
                enc_cuts = [t for t, _ in store.cuts if t.startswith("encoder.")]          # forward order
                ranges = _piece_ranges(store, [int(t.split(".")[1]) for t in enc_cuts])
                one = ops.const_scalar(total.device, 1.0) if total.dim() == 0 and total.dtype == torch.float32 \
                    else None                      # the root gradient: a constant, not a fill per step
                g_mem, g_y = torch.autograd.grad(total, [cuts["memory"], cuts["y_net"]], one, retain_graph=True)
                exchange(*ranges[0])
                up, g_up = cuts["memory"], g_mem
                for k, tag in enumerate(reversed(enc_cuts)):
                    (g_dn,) = torch.autograd.grad([up], [cuts[tag]], [g_up], retain_graph=True)
                    exchange(*ranges[1 + k])
                    up, g_up = cuts[tag], g_dn
                torch.autograd.backward([up, cuts["y_net"]], [g_up, g_y])
                exchange(*ranges[-1])
                for w in works:
                    if w is not None:
                        w.wait()
                solver._update(1.0 / max(solver.world_size, 1))
        finally:
            _WgradQueue.side = None
            store.collect_cuts = False
            store.cuts = []
        self._keep = (cuts, g_mem, g_y, up, g_up)          # the graph's buffers must outlive it
        self.templates = []
        for m, n in zip(metrics_objs, before):
            self.templates.append((m, m._pending[n:]))
            del m._pending[n:]

    def run(self, samples):
        from .metrics import _Deferred
        _copy_inputs(self.inputs, samples)
        self.graph.replay()
        self.solver.model.store.shadows_dirty = True
        for m, items in self.templates:
            for t in items:
                m._pending.append(_Deferred(t.owner, list(t.tensors), t.score).to_host())
        return self.loss, self.metrics


class _StreamHandle:
    """Handle of a collective enqueued on a side stream: ``wait`` makes the current stream wait for it."""

    def __init__(self, stream):
        self.stream = stream

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.stream)


def _encoder_cut_layers(store):
    """The encoder layers behind which the backward pass is cut (the rule of
    layers/transformer.py: thirds of a stack of three or more layers), from the parameter names."""
    n = 1 + max([int(s.name.split(".")[1]) for s in store.train_slots if s.name.startswith("encoder.")], default=-1)
    return [k for k in range(1, n) if n >= 3 and k % ((n + 2) // 3) == 0]


def _piece_ranges(store, enc_cut_layers):
    """Ranges of the flat gradient buffer that are final after each backward piece, in replay order:
    [decoder + heads], [encoder layers >= last cut], ..., [front end, y_net, first encoder layers].
    Relies on the registration order front end < encoder.0 < ... < decoder < heads; raises if the
    parameter layout does not have that shape."""
    n = store.n_train
    first = {}
    for s in store.train_slots:
        name = s.name
        if name.startswith("encoder."):
            key = int(name.split(".")[1])
        elif name.startswith(("x_net.", "y_net.")):
            key = -1
        else:
            key = 1 << 30
        first[key] = min(first.get(key, n), s.offset)
    bounds = [first.get(1 << 30, n)]
    for layer in sorted(enc_cut_layers, reverse=True):
        bounds.append(first[layer])
    bounds.append(0)
    # every slot must fall into the range its piece claims
    for s in store.train_slots:
        name = s.name
        if name.startswith("encoder."):
            li = int(name.split(".")[1])
            piece = 1 + sum(1 for c in sorted(enc_cut_layers, reverse=True) if li < c)
        elif name.startswith(("x_net.", "y_net.")):
            piece = len(bounds) - 1
        else:
            piece = 0
        hi = n if piece == 0 else bounds[piece - 1]
        if not (bounds[piece] <= s.offset < hi):
            raise RuntimeError("parameter layout does not allow a pipelined gradient exchange (%s)" % name)
    return [(bounds[i], n if i == 0 else bounds[i - 1]) for i in range(len(bounds))]


def bucket_ranges(seg_offsets, n_total, bucket_elems):
    """Cut the flat gradient buffer [0, n_total) into contiguous ranges of about ``bucket_elems``
    elements that end on variable boundaries, listed from the END of the buffer (the variables
    registered last receive their gradients first in the backward pass)."""
    bounds = sorted(set(int(o) for o in seg_offsets) | {0, int(n_total)})
    ranges = []
    end = int(n_total)
    idx = len(bounds) - 1
    while end > 0:
        start = end
        while idx > 0 and (end - start) < bucket_elems:
            idx -= 1
            start = bounds[idx]
        ranges.append((start, end))
        end = start
    return ranges


class HorovodSolver(BaseSolver):
    """Data-parallel solver: per-rank batches, gradients averaged over ranks before the clip
    (solver.py:142-218).  Launch with ``torchrun --nproc-per-node N``; the name is kept so that
    ``from athena import HorovodSolver`` keeps working."""

    # Bucket size of the pipelined exchange (pieces of the backward pass, each piece's gradients
    # exchanged while the next piece runs): small enough to start early.
    BUCKET_MB = float(os.environ.get("ATHENA_DP_BUCKET_MB", "25"))
    # Message size of the exchange behind the whole backward pass: nothing overlaps it, so fewer and
    # larger collectives are better (an xGMI ring is per-link bound and reaches its bandwidth only
    # on long messages); the AISHELL model's 84 MB go out as one all-reduce.
    EXCHANGE_MB = float(os.environ.get("ATHENA_DP_EXCHANGE_MB", "256"))

    def __init__(self, model, optimizer, sample_signature=None, eval_sample_signature=None, config=None, **kwargs):
        super().__init__(model, optimizer, sample_signature, eval_sample_signature, config, **kwargs)
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        st = model.store
        self._buckets = bucket_ranges(st.seg_offsets.tolist(), st.n_train, int(self.EXCHANGE_MB * (1 << 20) / 4))
        self._broadcast_done = False
        # ATHENA_FORCE_DP=1 runs the exchange (and the two-graph step) even with one rank: a way to
        # exercise the RCCL path on a single GPU
        self._force_exchange = os.environ.get("ATHENA_FORCE_DP", "0") == "1" and dist.is_initialized()
        # The solver's own RCCL communicator (utils/rccl.py): collectives issued through it can be
        # recorded in a HIP graph.  Created here, where every rank is at the same point, and checked
        # against the process group; None where it cannot be had (another backend, a failed check).
        self._rccl, self._side, self._plan = None, None, None
        if (self.world_size > 1 or self._force_exchange) and dist.is_initialized() and dist.get_backend() == "nccl" \
                and os.environ.get("ATHENA_DP_OWN_COMM", "1") != "0":
            from .utils.rccl import Communicator, all_ranks_agree
            comm, why = None, None
            try:
                comm = Communicator()          # collective: raises on every rank or on none
                comm.self_test()
            except Exception as e:      # noqa: BLE001 -- any failure means: collectives through torch.distributed
                why = e
            # which communicator carries the exchange is decided ONCE, here, and for all ranks together:
            # a rank whose self-test failed alone would otherwise issue its collectives on the process
            # group while the others issue theirs on their own communicators, and the job would hang
            if all_ranks_agree(why is None):
                self._rccl = comm
            else:
                if comm is not None:
                    comm.close()
                logging.warning("own RCCL communicator unavailable (%s): gradient exchange between graphs",
                                why or "failed on another rank")

    @staticmethod
    def initialize_devices(solver_gpus=None):
        """Join the process group (RANK / WORLD_SIZE / MASTER_* from the environment) and pin this
        process to its GPU (solver.py:146-160)."""
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if torch.cuda.is_available():
            idx = solver_gpus[local_rank] if solver_gpus else local_rank
            torch.cuda.set_device(idx)
        if not dist.is_initialized() and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # ATHENA_DP_BACKEND=gloo rehearses several ranks on fewer GPUs (the sums go through the host)
            backend = os.environ.get("ATHENA_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            dist.init_process_group(backend=backend)

    def broadcast_variables(self):
        """hvd.broadcast_variables(root_rank=0) of the model and optimizer state (solver.py:192-194)."""
        if self.world_size > 1:
            st = self.model.store
            for t in [st.params] + list(self.optimizer.variables()):
                if t.is_cuda and dist.get_backend() == "gloo":      # rehearsal backend: through the host
                    host = t.cpu()
                    dist.broadcast(host, src=0)
                    t.copy_(host)
                else:
                    dist.broadcast(t, src=0)
            st.shadows_dirty = True
        self._broadcast_done = True

    def _exchange_async(self, t):
        """Start the sum of a slice of the flat gradient buffer over the ranks; returns a handle whose
        ``wait`` makes the current stream wait for the result."""
        if self._rccl is not None:
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())      # the slice is final on the launch stream
            self._rccl.all_reduce_sum_(t, self._side)
            return _StreamHandle(self._side)
        if t.is_cuda and dist.get_backend() == "gloo":
            # rehearsal of the multi-rank step on a box with fewer GPUs than ranks: gloo sums on the host
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            t.copy_(host)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def _exchange_plan(self):
        """The slices of the flat gradient buffer in the order every kind of step exchanges them: the
        ranges that become final piece by piece in the backward pass (decoder and heads, encoder thirds
        from the top, front end), each in buckets of BUCKET_MB.  Ranks may be in different kinds of step
        at the same time (one replaying a recorded shape, another meeting a new shape eagerly): the
        collectives pair up only because all of them follow this one list."""
        if self._plan is None:
            st = self.model.store
            step = max(int(self.BUCKET_MB * (1 << 20) / 4), 1)
            try:
                ranges = _piece_ranges(st, _encoder_cut_layers(st))
            except (RuntimeError, KeyError, AttributeError, ValueError, IndexError):
                ranges = [(0, st.n_train)]      # a parameter layout without that shape: one range
            self._plan = [(lo, min(lo + step, b)) for a, b in ranges for lo in range(a, b, step)]
        return self._plan

    def _reduce_gradients(self):
        """Sum the flat gradient buffer over ranks, slice by slice of the exchange plan; the
        1/world_size of the average is folded into the optimizer kernel."""
        if self.world_size == 1 and not self._force_exchange:
            return 1.0
        g = self.model.store.grads
        works = [self._exchange_async(g[a:b]) for a, b in self._exchange_plan()]
        for w in works:
            if w is not None:
                w.wait()
        return 1.0 / self.world_size

    def _record(self, samples):
        # With more than one rank the exchange runs behind the backward pass, piece by piece.
        # ATHENA_DP_MODE picks how: "onegraph" (the default on RCCL: the pieces and their exchanges in
        # ONE HIP graph), "pipelined" (a chain of graphs with the collectives issued between them through
        # torch.distributed; what remains when the own communicator cannot be had), "split" (two graphs
        # around one exchange; also ATHENA_DP_PIPELINE=0).  With a single rank and no forced exchange
        # the plain single graph is recorded.
        # Ranks record at different times (their batch shapes differ), so the fallbacks below are local
        # decisions -- and may be: every kind of step sends the slices of _exchange_plan, in that order,
        # through _exchange_async, i.e. through the communicator chosen for all ranks in __init__.  A
        # rank that falls back changes how ITS step is cut into graphs, never what it exchanges or where.
        dp = self.world_size > 1 or self._force_exchange
        mode = os.environ.get("ATHENA_DP_MODE", "")
        if not mode:
            mode = "split" if os.environ.get("ATHENA_DP_PIPELINE", "1") == "0" else "onegraph"
        if dp and mode == "onegraph":
            if self._rccl is not None:
                try:
                    return _OneGraphPipelinedStep(self, samples)
                except RuntimeError as e:      # unusual parameter layout, a collective that cannot be recorded
                    logging.warning("one-graph gradient exchange disabled: %s", e)
            mode = "pipelined"
        if dp and mode == "pipelined":
            try:
                return _PipelinedGraphStep(self, samples)
            except RuntimeError as e:          # unusual parameter layout: exchange after the whole backward pass
                logging.warning("pipelined gradient exchange disabled: %s", e)
        return _GraphStep(self, samples)

    def train(self, dataset, total_batches=-1):
        """Update the model in 1 epoch (solver.py:176-197)."""
        for batch, samples in enumerate(dataset):
            if 0 <= total_batches <= batch:
                break
            samples = self.model.prepare_samples(samples)
            loss, metrics = self.train_step(samples)
            if batch == 0 and not self._broadcast_done:
                self.broadcast_variables()
            if batch % self.hparams.log_interval == 0:
                if self.rank == 0:
                    logging.info(self.metric_checker(loss, metrics))
                self.model.reset_metrics()


class DecoderSolver(BaseSolver):
    """Decoding driver (reference athena/solver.py:221-283): builds the beam-search decoder around the
    model's ``time_propagate`` and scores the predictions of a data set with the character accuracy."""
    default_config = {
        "inference_type": "asr",
        "decoder_type": "wfst_decoder",
        "model_avg_num": 1,
        "beam_size": 4,
        "ctc_weight": 0.0,
        "lm_weight": 0.1,
        "lm_type": "",
        "lm_path": None,
        "acoustic_scale": 10.0,
        "max_active": 80,
        "min_active": 0,
        "wfst_beam": 30.0,
        "max_seq_len": 100,
        "wfst_graph": None
    }

    def __init__(self, model, data_descriptions=None, config=None):
        from .tools.beam_search import BeamSearchDecoder
        from .utils.hparam import register_and_parse_hparams
        self.model = model
        self.optimizer = None
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        if self.hparams.decoder_type == "beam_search_decoder":
            self.decoder = BeamSearchDecoder.build_decoder(self.hparams, model.num_class, model.sos, model.eos,
                                                           model.time_propagate)
        elif self.hparams.decoder_type == "wfst_decoder":
            raise NotImplementedError("the WFST decoder (athena/tools/wfst_decoder.py) is not built")
        else:
            raise ValueError("This decoder type is not supported")

    def inference(self, dataset, rank_size=1):
        """decode the model: one utterance per batch; logs predictions, labels and the running accuracy"""
        import time
        from .metrics import CharactorAccuracy
        from .utils.misc import validate_seqs
        if dataset is None:
            return None
        metric = CharactorAccuracy(rank_size=rank_size)
        for samples in dataset:
            begin = time.time()
            samples = self.model.prepare_samples(samples)
            predictions = self.model.decode(samples, self.hparams, self.decoder)
            validated = validate_seqs(predictions, self.model.eos)[0]
            num_errs, _ = metric.update_state(validated, samples)
            logging.info("predictions: %s\tlabels: %s\terrs: %d\tavg_acc: %.4f\tsec/iter: %.4f",
                         predictions.tolist(), samples["output"].tolist(), num_errs, metric.result(),
                         time.time() - begin)
        logging.info("decoding finished")
        return metric.result()
