"""Data-parallel gradient exchange for the flat ParamStore: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no explicit collective (Lightning's implicit DDP would all-reduce ~440 tensors in 25 MB
buckets, train.py:93-98).  Here the gradient buffer is laid out in the order backward completes it
(param_store.SEGMENTS: RGB decoder 186 MB, range-view decoder 135 MB, voxel decoder 12 MB, policy 24 MB, RSSM
53 MB, fusion ~210 MB, the two encoder branches ~57 MB each at base_1d), so each segment is ONE contiguous
sum-all-reduce, issued on a side HIP stream the moment autograd leaves the segment and overlapped with the rest
of backward: the first 333 MB are in flight while the RGB decoder (the longest stretch of backward) still runs, and
only the image-encoder segment is exposed after backward.  xGMI is point-to-point, so few large messages keep all 7
links per GPU busy; BatchNorm statistics stay per-GPU (reference: sync_batchnorm commented out, train.py:98).  The
1/world_size averaging is folded into the fused AdamW kernel (grad_scale)."""
import torch
import torch.distributed as dist


class SegmentedGradReducer:
    def __init__(self, store, group=None, overlap=True, force_collectives=False, verify=False, fake_peers=0):
        """force_collectives: issue the all-reduces even in a one-rank group (exercises the RCCL / side-stream path on a
        single GPU; `bench.py` sets it when MUVO_BENCH_FORCE_DIST=1).
        verify: keep a copy of every segment as it is handed to the collective and compare it in finish() with what the
        segment holds once backward has ended (one-rank groups only: a sum over one rank changes nothing) — proves that
        no kernel wrote into a segment after its hook fired (tests/test_dp_gpu.py).
        fake_peers = G > 1 (MUVO_DP_FAKE_PEERS on a one-GPU box, no process group needed): every segment's all-reduce is
        replaced by ops.fake_allreduce on the communication stream - a kernel with a collective's local footprint (a few
        resident workgroups, the segment read twice and written once at the xGMI ring's rate, values unchanged) - so the
        overlap with the big-LDS convolution tiles and the persistent recurrent kernels is MEASURED with `exposed_ms` per
        segment instead of being a no-op (a one-rank RCCL all-reduce moves no byte)."""
        self.store = store
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.fake_peers = int(fake_peers) if (int(fake_peers) > 1 and self.world == 1) else 0
        self.force = (bool(force_collectives) and dist.is_initialized()) or self.fake_peers > 0
        self.overlap = overlap
        self.verify = verify
        self.accumulating = False     # True while backward runs for a non-final micro-batch of gradient accumulation
        self.is_cuda = store.flat_grad.is_cuda
        self.side = torch.cuda.Stream() if (self.is_cuda and (self.world > 1 or self.force)) else None
        if self.side is not None:
            # the communication stream and RCCL's own stream are two of the four streams the chip runs well side by side
            # (muvo_amd/ops.py: side_stream; the communication stream itself only carries waits): the model folds its
            # branches onto TWO side streams from now on
            import os
            from muvo_amd import ops
            if 'MUVO_STREAM_BUDGET' not in os.environ:
                ops.set_stream_budget(2)
        self._done = []
        self._handles = []
        self._snap = {}
        self._begun = False
        self.ranges = {name: (a, b) for name, a, b in store.segment_ranges}
        self.order = [name for name, _, _ in store.segment_ranges]
        from muvo_amd.param_store import SEGMENTS
        self._rank = {name: i for i, (name, _) in enumerate(SEGMENTS)}     # completion order, also of absent segments
        # parameters per segment: a gradient tensor that is not the flat view (installed by AccumulateGrad after something set
        # p.grad to None) is copied into its slot BEFORE the segment is exchanged - afterwards it would never be reduced
        self._seg_params = {name: [p for p in store.params if a <= store._off[id(p)] < b] for name, (a, b) in self.ranges.items()}
        self.launch_log = []          # (segment, was launched from a backward hook) of the last step
        self.late_writes = {}         # verify: segment -> max |difference|
        # timing (bench.py, MUVO_DP_TIMING=1): HIP events around every segment's collective on the side stream and at the
        # join, so that a multi-GPU run says by itself how long each all-reduce took and how much of the exchange was NOT
        # hidden behind backward (the time the optimizer stream waits in finish())
        import os
        self.timing = os.environ.get('MUVO_DP_TIMING') == '1'
        self._ev = []                 # (segment, start event, end event) of the current step
        self._steps_timed = []        # per finished step: (events of the segments, join event on the main stream, last side event)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def begin_step(self):
        """Start of an optimisation step (before backward).  Optional since finish() resets the state, but calling it
        twice without a finish() in between means a backward ended without its exchange: refuse."""
        if self._begun and self._done:
            raise RuntimeError('SegmentedGradReducer.begin_step(): the previous backward was never finish()ed '
                               f'(segments already sent: {self._done})')
        self._done, self._handles, self._snap = [], [], {}
        self._begun = True

    def _launch(self, name, from_hook):
        if name in self._done or name not in self.ranges:
            return
        if self.timing:
            import time
            t0 = time.perf_counter()
            try:
                return self._launch_impl(name, from_hook)
            finally:
                self.host_s = getattr(self, 'host_s', 0.0) + time.perf_counter() - t0
        return self._launch_impl(name, from_hook)

    def _launch_impl(self, name, from_hook):
        self._done.append(name)
        self.launch_log.append((name, from_hook))
        if self.world == 1 and not self.force:
            return
        a, b = self.ranges[name]
        self.store.settle_grads(self._seg_params[name])
        buf = self.store.flat_grad[a:b]
        if self.verify:
            from muvo_amd import ops
            ops.join_side_streams()
            self._snap[name] = buf.clone()
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            from muvo_amd import ops
            ops.join_side_streams(into=self.side)      # the segment's kernels may have run on a branch stream (ops.branch)
            with torch.cuda.stream(self.side):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.side)
                if self.fake_peers:
                    from muvo_amd import ops
                    ops.fake_allreduce(buf, self.fake_peers)
                else:
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                if self.timing:
                    e1.record(self.side)
                    self._ev.append((name, from_hook, e0, e1, (b - a) * 4))
        else:
            self._handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def segment_done(self, name):
        """autograd hook: backward has finished every parameter of `name` (and of all earlier segments)."""
        if not self.overlap or self.accumulating:
            return
        if not self._done:
            self.launch_log = []
        upto = self._rank[name]
        for n in self.order:
            if self._rank[n] > upto:
                break
            self._launch(n, True)

    def skip(self):
        """End of a backward whose gradients stay local (gradient accumulation: only the last micro-batch exchanges).
        Hooks must not have sent anything: set `overlap=False` / `accumulating=True` before such a backward."""
        if self._done:
            raise RuntimeError('SegmentedGradReducer.skip(): segments were already sent during this backward; set '
                               '`reducer.accumulating = True` before the backward of a non-final micro-batch')
        self._begun = False

    def finish(self):
        """after backward: reduce whatever is left and make the optimizer stream wait for the exchange."""
        if not self._done:
            self.launch_log = []
        for n in self.order:
            self._launch(n, False)
        for h in self._handles:
            h.wait()
        if self.side is not None:
            if self.timing and self._ev:
                join = torch.cuda.Event(enable_timing=True)
                join.record(torch.cuda.current_stream())        # where backward ended on the compute stream
                self._steps_timed.append((self._ev, join))
                self._ev = []
                if len(self._steps_timed) > 64:
                    self._steps_timed.pop(0)
            torch.cuda.current_stream().wait_stream(self.side)
        assert self._done == self.order, (self._done, self.order)   # every segment exactly once, in layout order
        if self.verify and self.world == 1:
            self.late_writes = {}
            for name, snap in self._snap.items():
                a, b = self.ranges[name]
                self.late_writes[name] = float((self.store.flat_grad[a:b] - snap).abs().max())
        self._done, self._handles, self._snap = [], [], {}
        self._begun = False

    def timing_report(self):
        """Per segment (mean over the timed steps): duration of its all-reduce on the side stream, achieved bus bandwidth
        (algorithm bytes 2 (G-1)/G x size), whether a backward hook launched it, and `exposed_ms`: how long after the end of
        backward on the compute stream the segment's collective finished (<= 0: fully hidden).  The step's exposed exchange
        time is the maximum over its segments.  Synchronises; call it after the timed region."""
        if not self._steps_timed:
            return None
        torch.cuda.synchronize()
        acc, exposed = {}, []
        for evs, join in self._steps_timed:
            worst = 0.0
            for name, from_hook, e0, e1, nbytes in evs:
                a = acc.setdefault(name, dict(ms=0.0, exposed_ms=0.0, n=0, bytes=nbytes, from_hook=from_hook))
                ms = e0.elapsed_time(e1)
                ex = join.elapsed_time(e1)                      # negative: done before backward ended
                a['ms'] += ms
                a['exposed_ms'] += ex
                a['n'] += 1
                worst = max(worst, ex)
            exposed.append(worst)
        g = self.fake_peers or self.world
        segs = {}
        for name, a in acc.items():
            ms = a['ms'] / a['n']
            segs[name] = dict(mbytes=round(a['bytes'] / 2 ** 20, 1), allreduce_ms=round(ms, 3),
                              busbw_gbs=round(2.0 * (g - 1) / max(g, 1) * a['bytes'] / max(ms, 1e-6) / 1e6, 1),
                              from_hook=a['from_hook'], exposed_ms=round(a['exposed_ms'] / a['n'], 3))
        self._steps_timed = []
        host_ms = round(getattr(self, 'host_s', 0.0) * 1e3 / max(len(exposed), 1), 3)     # host time inside the collective calls
        self.host_s = 0.0
        return dict(steps=len(exposed), exposed_ms_per_step=round(sum(exposed) / len(exposed), 3), host_ms_per_step=host_ms, segments=segs)
