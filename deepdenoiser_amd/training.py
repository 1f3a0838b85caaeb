"""Training driver for the hot path: the MI355X-native counterpart of the reference's `model_fn` train op
(`tf.train.AdamOptimizer(lr).minimize(loss)`, reference TensorFlow/Training.py:700-702) plus NEW data parallelism
(the reference is single-device).

One process per GPU.  Every rank holds a full replica (identical seed); the mini-batch of render-pass tiles is sharded
across ranks (per-rank data); gradients live in ONE flat fp32 arena in variable-creation order, so all-reduce buckets
are zero-copy slices of it.  Backward runs in reverse creation order => the arena becomes final from its TAIL; the step
is cut into hipGraph segments at bucket boundaries and each bucket's RCCL all-reduce is issued on a side stream as soon
as its segment finishes, overlapping the remaining backward.  The payload is small (~6.8 MB fp32 for the flagship
config) and latency-bound on xGMI, so the bucket count is kept small (default 4).
"""
import torch

from . import _lib as L


def plan_buckets(params, last_writer, n_ops, total, n_buckets):
    """Cut a reverse program of `n_ops` ops into `n_buckets` segments by gradient readiness.  Pure integer logic (CPU-testable).

    params:      [(name, offset, size)] layout of the flat gradient arena (variable-creation order)
    last_writer: {name: index of the LAST op of the reverse program that writes this parameter's gradient}
    Returns [(op_begin, op_end, lo, hi)]: after ops [op_begin, op_end) have run, arena slice [lo, hi) is final and may be
    all-reduced while later segments run.  Segments tile [0, n_ops) in order; slices tile [0, total), taken from the TAIL of
    the arena first (backward runs in reverse creation order, so the tail is ready first)."""
    n_buckets = max(1, min(n_buckets, total))
    edges = [total * k // n_buckets for k in range(n_buckets + 1)]
    out, begin, done = [], 0, 0
    for k in reversed(range(n_buckets)):
        lo, hi = edges[k], edges[k + 1]
        ready = 0
        for name, off, size in params:
            if off < hi and off + size > lo:
                ready = max(ready, last_writer.get(name, -1) + 1)
        done = max(done, ready)
        out.append([begin, done, lo, hi])
        begin = done
    out[-1][1] = n_ops                      # trailing ops that write no parameter gradient (input-side glue)
    return [tuple(x) for x in out]


class GradientReducer:
    """Sum all-reduce of slices of the flat gradient arena, issued asynchronously (side HIP stream on GPU, async work handles on
    CPU/gloo) so that it overlaps the rest of backward.  `wait()` joins everything before the optimizer reads the arena."""

    def __init__(self, grads, world_size, group=None, force=False):
        """force: issue the collectives even in a world of one rank (a one-rank RCCL communicator still runs its all-reduce kernel on the
        side stream: how the RCCL code path is exercised on a one-GPU box, tests/test_gpu_rccl.py)."""
        self.grads, self.world, self.group = grads, world_size, group
        self.active = world_size > 1 or force
        self._stream = torch.cuda.Stream(device=grads.device) if (self.active and grads.is_cuda) else None
        self._works = []
        self.timing = None             # measure_collectives(): list of (start, end) events per all-reduce of the current step

    def launch(self, lo, hi):
        if not self.active or hi <= lo:
            return
        import torch.distributed as dist
        view = self.grads[lo:hi]
        if self._stream is not None:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self._stream):
                self._stream.wait_event(ev)
                if self.timing is not None:
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record()
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                if self.timing is not None:
                    t1.record()
                    self.timing.append((t0, t1))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        if self._stream is not None:
            if self.timing is not None:       # when the launch stream gets here (everything it had to do beside the collectives is queued before)
                self._ready = torch.cuda.Event(enable_timing=True)
                self._ready.record()
            torch.cuda.current_stream().wait_stream(self._stream)
        for w in self._works:
            w.wait()
        self._works = []

    @property
    def grad_scale(self):
        """Factor the optimizer applies to the summed gradients (mean over ranks; every rank's loss is a mean over its shard)."""
        return 1.0 / self.world


class Trainer:
    def __init__(self, arch, training_json, B, H, W, world_size=1, use_graph=True, n_buckets=4, force_segments=False, force_collectives=False):
        self.arch, self.world = arch, world_size
        self.program = arch.program(B, H, W, training_json=training_json)
        self.use_graph = use_graph
        self.n_buckets = max(1, n_buckets) if (world_size > 1 or force_segments or force_collectives) else 1
        self._segments = None          # list of (ops, arena slice)
        self._graphs = None
        self.reducer = GradientReducer(arch.params.grads, world_size, force=force_collectives)
        self._warm = 0
        # Masked means divide by the batch-GLOBAL mask count (Training.py:131-137).  With the batch sharded over ranks the per-rank counts
        # are summed (one all-reduce of a few dozen floats per step, before the forward) and divided by the world size: the optimizer
        # averages the ranks' gradients, so a rank's masked term must be  sum_rank(d * mask) * world / count_global.
        self._reduce_masks = (world_size > 1 or force_collectives) and self.program.masked
        if self._reduce_masks:
            self.program.mask_reduce = self._mask_reduce

    # ------------------------------------------------------------------ segmentation by gradient readiness
    def _build_segments(self):
        prog, ps = self.program, self.arch.params
        g = prog.g
        # label-only launches (scaled targets, mask counts) join segment 0 unless the mask counts need their all-reduce first
        head = list(g.pack_ops) + ([] if self._reduce_masks else list(prog.label_ops)) + list(g.fwd_ops)
        bwd = list(g.bwd_ops)
        last_writer = {}
        for i, op in enumerate(bwd):
            for p in getattr(op, "grad_params", ()):
                last_writer[p.name] = i
        plan = plan_buckets([(p.name, p.offset, p.size) for p in ps.params], last_writer, len(bwd), ps.values.numel(), self.n_buckets)
        self._segments = [((head if k == 0 else []) + bwd[b:e], (lo, hi)) for k, (b, e, lo, hi) in enumerate(plan)]

    def _run_segment_eager(self, idx):
        ops, _ = self._segments[idx]
        if idx == 0:
            self.program.zero_grads()
        self.program.g.run(ops)

    def _capture(self):
        self._graphs = []
        for idx in range(len(self._segments)):
            gr = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of torch.distributed queries events while we capture; in the default
            # "global" mode any such call from another thread invalidates the capture
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                self._run_segment_eager(idx)
            self._graphs.append(gr)

    def _mask_reduce(self, mask_sums):
        import torch.distributed as dist
        dist.all_reduce(mask_sums, op=dist.ReduceOp.SUM, group=self.reducer.group)
        mask_sums.mul_(1.0 / self.world)

    # ------------------------------------------------------------------ one optimisation step
    def step(self):
        prog = self.program
        if self._segments is None:
            self._build_segments()
        if self._reduce_masks:
            prog.run_label_ops()              # eager, outside the hipGraphs: contains a collective
        if self.use_graph and self._graphs is None and self._warm >= 2:
            torch.cuda.synchronize()
            self._capture()
        self._warm += 1
        for idx, (ops, sl) in enumerate(self._segments):
            if self._graphs is not None:
                self._graphs[idx].replay()
            else:
                self._run_segment_eager(idx)
            self.reducer.launch(*sl)          # RCCL all-reduce of the slice that just became final (no-op for world 1)
        self.reducer.wait()
        prog.adam(grad_scale=self.reducer.grad_scale)
        return prog.loss_buf

    def measure_collectives(self, steps=5):
        """Per-step time of the gradient all-reduces on the side stream and the part of it the launch stream had to wait for (HIP events; run
        OUTSIDE a timed region, by every rank).  exposed = (end of the last all-reduce) - (the launch stream reaching the join), clipped at 0:
        what a scaling curve below the ideal loses to communication as opposed to launch overhead or input feeding.  None without a side
        stream (one rank without forced collectives, CPU / gloo work handles)."""
        r = self.reducer
        if r._stream is None:
            return None
        tot, exposed, n = 0.0, 0.0, 0
        for _ in range(steps):
            r.timing = []
            self.step()
            torch.cuda.synchronize()
            tot += sum(a.elapsed_time(b) for a, b in r.timing)
            n = len(r.timing)
            if r.timing:
                exposed += max(0.0, r._ready.elapsed_time(r.timing[-1][1]))
        r.timing = None
        return {"allreduce_ms_per_step": tot / steps, "exposed_ms_per_step": exposed / steps, "allreduces_per_step": n,
                "bytes_per_step": int(self.arch.params.grads.numel() * self.arch.params.grads.element_size())}
