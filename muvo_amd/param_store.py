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

# reverse-execution order of the base_1d model: backward completes these prefixes top to bottom
SEGMENTS = (
    ('decoders', ('voxel_decoder.', 'lidar_re.', 'rgb_decoder.', 'policy.', 'lidar_segmentation.', 'sem_image_decoder.',
                  'depth_image_decoder.', 'bev_decoder.')),            # done when d(state) arrives
    ('rssm', ('rssm.',)),                                                                  # done when d(embedding) arrives
    ('fusion', ('features_combine.', 'speed_enc.', 'backbone_route.', 'image_feature_conv.', 'lidar_feature_conv.',
                'transformer_encoder.')),                                                  # done when d(tokens) arrives
    ('encoders', ('type_embedding', 'feat_decoder.', 'range_view_decoder.', 'encoder.', 'range_view_encoder.')),
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
        self.nodecay_params = [p for n, p, _ in order if p.dim() == 1 or any(s in n for s in skip_decay)]
        self.decay_params = [p for n, p, _ in order if not (p.dim() == 1 or any(s in n for s in skip_decay))]

    def zero_grad(self):
        self.flat_grad.zero_()
        for _, p in self.unused:
            p.grad = None

    def rebind_grads(self):
        """Re-attach p.grad views (after something set them to None, e.g. optimizer.zero_grad(set_to_none=True))."""
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * self._off[id(p)]:
                o = self._off[id(p)]
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)
