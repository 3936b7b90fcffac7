"""Flat parameter / gradient / optimizer-state storage laid out for the MI355X step.

All trainable tensors live in ONE fp32 buffer (and their gradients in another), ordered by the segments in
which backward finishes them (decoders first, encoders last); inside a segment the reference's two weight-decay
groups (trainer.py:1031-1051: 1-D tensors -> no decay) are contiguous.  Consequences:
  * zero_grad is one memset, AdamW is <= 2 kernel launches per segment instead of 452,
  * the data-parallel gradient exchange is one large RCCL all-reduce per segment, launched on a side stream as
    soon as backward leaves that segment (muvo_amd/parallel.py), instead of 440 small ones.
Parameters the reference registers but never uses (`encoder_layer.*`) keep their own storage,
never get a gradient and are skipped by the optimizer, exactly like torch.optim.AdamW skips `grad is None`."""
import torch

# Order in which backward finishes the parameters of the model (autograd runs the most recently recorded forward nodes
# first: the reference, mile.py:437-487, calls policy, rgb, lidar_re, the config-off heads and the voxel decoder in that order;
# muvo_amd/models/mile.py records voxel, range-view, RGB decoder - see the comment there - so backward completes RGB, range-view, voxel).  Each entry is one contiguous range of the flat gradient buffer = ONE all-reduce, launched
# by the hook named in the comment (muvo_amd/models/mile.py `_mark` / `_hook`).
SEGMENTS = (
    ('depth_image_decoder', ('depth_image_decoder.',)),
    ('sem_image_decoder', ('sem_image_decoder.',)),
    ('lidar_segmentation', ('lidar_segmentation.',)),
    ('bev_decoder', ('bev_decoder.',)),
    ('rgb_decoder', ('rgb_decoder.',)),
    ('lidar_re', ('lidar_re.',)),
    ('voxel_decoder', ('voxel_decoder.',)),                      # recorded first in forward (models/mile.py), so finished last of the decoders
    ('policy', ('policy.',)),                                    # d(state) complete (every consumer of the state done)
    ('rssm', ('rssm.',)),                                        # d(embedding) arrives
    ('fusion', ('features_combine.', 'speed_enc.', 'backbone_route.', 'image_feature_conv.', 'lidar_feature_conv.',
                'transformer_encoder.')),                        # d(tokens) arrives
    ('lidar_branch', ('type_embedding', 'range_view_decoder.', 'range_view_encoder.')),   # image branch about to start
    ('image_branch', ('feat_decoder.', 'depth_decoder.', 'depth.', 'bev_down_sample_4.', 'frustum_pooling.', 'encoder.')),
)
UNUSED_PREFIXES = ('encoder_layer.',)


class ParamStore:
    def __init__(self, model: torch.nn.Module, skip_decay=('relative_position_bias_table',)):
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.unused = [(n, p) for n, p in named if n.startswith(UNUSED_PREFIXES)]
        used = [(n, p) for n, p in named if not n.startswith(UNUSED_PREFIXES)]
        seg_of = {}
        for n, p in used:
            for si, (_, prefixes) in enumerate(SEGMENTS):
                if n.startswith(prefixes):
                    seg_of[n] = si
                    break
            else:
                seg_of[n] = len(SEGMENTS) - 1
        order, self.ranges = [], []   # ranges: (segment index, decay flag, start, end)
        off = 0
        for si in range(len(SEGMENTS)):
            for decay in (False, True):
                start = off
                for n, p in used:
                    is_nodecay = p.dim() == 1 or any(s in n for s in skip_decay)
                    if seg_of[n] == si and (not is_nodecay) == decay:
                        off = (off + 3) & ~3       # 16-byte aligned tensors: the kernels read weights as float4
                        order.append((n, p, off))
                        off += p.numel()
                if off > start:
                    self.ranges.append((si, decay, start, off))
        self.numel = off
        dev = used[0][1].device
        self.flat_param = torch.zeros(off, device=dev, dtype=torch.float32)   # alignment gaps stay zero
        self.flat_grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        self.offsets = {}
        with torch.no_grad():
            for n, p, o in order:
                view = self.flat_param[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
                self.offsets[n] = (o, p.numel())
        self.segment_ranges = []
        for si in range(len(SEGMENTS)):
            rs = [r for r in self.ranges if r[0] == si]
            if rs:
                self.segment_ranges.append((SEGMENTS[si][0], min(r[2] for r in rs), max(r[3] for r in rs)))
        self.params = [p for _, p, _ in order]
        self._off = {id(p): o for _, p, o in order}
        # optimizer param groups in the REFERENCE's order (trainer.py:1031-1051 walks model.named_parameters()), so a
        # torch.optim.AdamW state_dict of the reference maps index -> parameter identically (muvo_amd/optim.py)
        nodecay = lambda n, p: p.dim() == 1 or any(s in n for s in skip_decay)
        self.nodecay_params = [p for n, p in named if nodecay(n, p)]
        self.decay_params = [p for n, p in named if not nodecay(n, p)]
        self.used_ids = {id(p) for p in self.params}
        for n, p, o in order:
            p._muvo_flat_grad = self.flat_grad[o:o + p.numel()].view(p.shape)   # ops.grad_of re-binds to it

    def zero_grad(self):
        """All gradients zero, every p.grad the flat view again.  A foreign gradient tensor (installed by AccumulateGrad after
        `p.grad = None`, or assigned from outside) is DISCARDED here, like torch's zero_grad discards it: copying it back would
        put the old gradient into the freshly zeroed slot."""
        self.flat_grad.zero_()
        self.rebind_grads(copy_foreign=False)
        for _, p in self.unused:
            p.grad = None

    def rebind_grads(self, copy_foreign=True):
        """Re-attach the p.grad views after something replaced them (nn.Module.zero_grad() / `p.grad = None` set them to
        None; a kernel that found None re-binds through ops.grad_of).  A gradient tensor that is NOT the flat view (assigned
        from outside) is copied into its slot first, so its contents are not lost (copy_foreign; zero_grad passes False)."""
        base = self.flat_grad.data_ptr()
        for p in self.params:
            view = p._muvo_flat_grad
            g = p.grad
            if g is None:
                p.grad = view
            elif g.data_ptr() != base + 4 * self._off[id(p)]:
                if copy_foreign:
                    view.copy_(g)
                p.grad = view

    def settle_grads(self, params=None):
        """Before the optimizer (or, per segment, the gradient exchange: parallel.SegmentedGradReducer) reads flat_grad: a
        parameter whose .grad is None got no gradient since something reset it (its slot may hold the previous step's values)
        -> zero the slot; foreign gradient tensors are copied in."""
        base = self.flat_grad.data_ptr()
        for p in (self.params if params is None else params):
            g = p.grad
            if g is None:
                p._muvo_flat_grad.zero_()
                p.grad = p._muvo_flat_grad
            elif g.data_ptr() != base + 4 * self._off[id(p)]:
                p._muvo_flat_grad.copy_(g)
                p.grad = p._muvo_flat_grad
