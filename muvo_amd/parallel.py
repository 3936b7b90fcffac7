"""Data-parallel gradient exchange for the flat ParamStore: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no explicit collective (Lightning's implicit DDP would all-reduce ~440 tensors in 25 MB
buckets).  Here the gradient buffer is laid out in the order backward completes it (param_store.SEGMENTS), so
each segment is ONE contiguous sum-all-reduce (357 MB decoders, 53 MB RSSM, ~210 MB fusion, ~75 MB encoders at
base_1d), issued on a side HIP stream the moment autograd leaves the segment and overlapped with the rest of
backward.  xGMI is point-to-point, so few large messages keep all 7 links per GPU busy; BatchNorm statistics stay
per-GPU (reference: sync_batchnorm commented out, train.py:98).  The 1/world_size averaging is folded into the
fused AdamW kernel (grad_scale)."""
import torch
import torch.distributed as dist


class SegmentedGradReducer:
    def __init__(self, store, group=None, overlap=True, force_collectives=False):
        """force_collectives: issue the all-reduces even in a one-rank group (exercises the RCCL / side-stream path on a
        single GPU; `bench.py` sets it when MUVO_BENCH_FORCE_DIST=1)."""
        self.store = store
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force_collectives) and dist.is_initialized()
        self.overlap = overlap
        self.is_cuda = store.flat_grad.is_cuda
        self.side = torch.cuda.Stream() if (self.is_cuda and (self.world > 1 or self.force)) else None
        self._done = set()
        self._handles = []
        self.ranges = {name: (a, b) for name, a, b in store.segment_ranges}
        self.order = [name for name, _, _ in store.segment_ranges]

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def begin_step(self):
        self._done.clear()
        self._handles.clear()

    def _launch(self, name):
        if name in self._done or name not in self.ranges:
            return
        self._done.add(name)
        if self.world == 1 and not self.force:
            return
        a, b = self.ranges[name]
        buf = self.store.flat_grad[a:b]
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self._handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def segment_done(self, name):
        """autograd hook: backward has finished every parameter of `name` (and of all earlier segments)."""
        if not self.overlap:
            return
        for n in self.order:
            self._launch(n)
            if n == name:
                break

    def finish(self):
        """after backward: reduce whatever is left and make the optimizer stream wait for the exchange."""
        for n in self.order:
            self._launch(n)
        for h in self._handles:
            h.wait()
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
