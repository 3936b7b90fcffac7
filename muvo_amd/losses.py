"""Loss modules with the reference's names/signatures (muvo/losses.py:53-287) on the fused HIP loss kernels.
`WorldModelTrainer.compute_loss` uses the fused multi-part entry points directly; these classes exist so code
written against the reference's loss objects keeps working."""
import torch
import torch.nn as nn

from muvo_amd import ops

VOXEL_SEG_WEIGHTS = (1.0, 1.0, 1.0, 1.5, 2.0, 3.0, 1.0, 1.0, 1.0)      # constants.py:39
SEMANTIC_SEG_WEIGHTS = (1.0, 1.0, 1.0, 2.0, 3.0, 1.0, 1.0, 1.0)        # constants.py:33


class RegressionLoss(nn.Module):
    """muvo/losses.py:53-71: |d| (norm 1) or d^2 (norm 2) summed over the channel dimension, mean over the rest."""

    def __init__(self, norm, channel_dim=-1):
        super().__init__()
        if norm not in (1, 2):
            raise ValueError(f'Expected norm 1 or 2, but got norm={norm}')
        if channel_dim != -1:
            raise NotImplementedError('the reference only ever uses the last dimension (trainer.py:57)')
        self.norm = norm

    def forward(self, prediction, target):
        if self.norm == 1:
            return ops.l1_rows_loss(prediction, target, 1.0)[0]
        # norm 2 on the masked spatial-loss kernel: rows = "frames" of one pixel, every row counted (all-ones mask)
        rows = prediction.numel() // prediction.shape[-1]
        p = prediction.reshape(1, rows, prediction.shape[-1], 1, 1)
        t = target.reshape(1, rows, prediction.shape[-1], 1, 1).float()
        ones = torch.ones(1, rows, 1, 1, 1, device=p.device, dtype=torch.uint8)
        return ops.spatial_losses(p, t, [(0, prediction.shape[-1], 2, 1.0)], mask=ones)[0]


class SpatialRegressionLoss(nn.Module):
    def __init__(self, norm, ignore_index=255):
        super().__init__()
        if norm not in (1, 2):
            raise ValueError(f'Expected norm 1 or 2, but got norm={norm}')
        self.norm, self.ignore_index = norm, ignore_index

    def forward(self, prediction, target, instance_mask=None):
        assert prediction.dim() == 5, 'Must be a 5D tensor'
        c = prediction.shape[2]
        return ops.spatial_losses(prediction, target, [(0, c, self.norm, 1.0)], float(self.ignore_index), mask=instance_mask)[0]


class KLLoss(nn.Module):
    def __init__(self, alpha):
        super().__init__()
        self.alpha = alpha

    def forward(self, prior, posterior):
        return ops.kl_loss(prior['mu'], prior['sigma'], posterior['mu'], posterior['sigma'], 1.0, self.alpha)[0]


class _VoxelTriple(nn.Module):
    index = 0

    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, prediction, target):
        return ops.voxel_losses(prediction, target, 1.0)[self.index]


class VoxelLoss(_VoxelTriple):
    """muvo/losses.py:144-186.  Plain / class-weighted mean: the fused voxel-loss kernel.  Top-k: the per-voxel (weighted) cross
    entropy map from the segmentation kernels, `topk` over every frame's voxels (a selection), mean."""
    index = 0

    def __init__(self, use_top_k=False, top_k_ratio=1.0, use_weights=False, poly_one=False, poly_one_coefficient=0.0):
        super().__init__()
        if poly_one:
            raise NotImplementedError('poly-1 is never enabled by the reference trainer (trainer.py:163-170)')
        self.use_top_k, self.top_k_ratio = use_top_k, top_k_ratio
        self.weights = VOXEL_SEG_WEIGHTS if use_weights else None

    def forward(self, prediction, target):
        cw = torch.tensor(self.weights, dtype=torch.float32, device=prediction.device) if self.weights is not None else None
        if cw is not None and prediction.shape[2] != len(self.weights):
            raise ValueError(f'VOXEL_SEG.USE_WEIGHTS needs the {len(self.weights)}-class head (constants.py:39), got {prediction.shape[2]} classes')
        if not self.use_top_k:
            return ops.voxel_losses(prediction, target, 1.0, cw)[0]
        b, s, c = prediction.shape[:3]
        loss = ops.seg_ce_pixel_loss(prediction.reshape(b * s, c, -1), target.reshape(b * s, -1), cw).view(b, s, -1)
        k = int(self.top_k_ratio * loss.shape[2])
        return torch.mean(loss.topk(k, dim=-1)[0])


class SemScalLoss(_VoxelTriple):
    index = 1


class GeoScalLoss(_VoxelTriple):
    index = 2




class SegmentationLoss(nn.Module):
    """muvo/losses.py:9-50: per-pixel (optionally class-weighted) cross entropy on the HIP kernel, then the mean of the
    top-k hardest pixels of every frame (k = int(ratio * h * w)) or of all pixels.  The top-k selection itself is
    torch.topk on the per-pixel loss (a selection, no arithmetic); its gradient flows back into the kernel's backward."""

    def __init__(self, use_top_k=False, top_k_ratio=1.0, use_weights=False, poly_one=False, poly_one_coefficient=0.0, is_bev=True):
        super().__init__()
        if poly_one:
            raise NotImplementedError('poly-1 is never enabled by the reference trainer (trainer.py:132-161)')
        self.use_top_k, self.top_k_ratio, self.use_weights = use_top_k, top_k_ratio, use_weights
        self.weights = (SEMANTIC_SEG_WEIGHTS if is_bev else VOXEL_SEG_WEIGHTS) if use_weights else None

    def forward(self, prediction, target):
        b, s, c, h, w = prediction.shape
        cw = torch.tensor(self.weights, dtype=torch.float32, device=prediction.device) if self.weights is not None else None
        loss = ops.seg_ce_pixel_loss(prediction.reshape(b * s, c, h, w), target.reshape(b * s, h, w), cw).view(b, s, -1)
        if self.use_top_k:
            k = int(self.top_k_ratio * loss.shape[2])
            loss = loss.topk(k, dim=-1)[0]
        return torch.mean(loss)


class _SSIMMeanFn(torch.autograd.Function):
    """mean over frames of the per-frame mean SSIM map (SSIMLoss.forward, losses.py:341-348), differentiable in the prediction."""

    @staticmethod
    def forward(ctx, prediction, target, win, c1, c2):
        n, c, h, w = prediction.shape
        p, t = prediction.contiguous(), target.float().contiguous()
        oh, ow = h - 10, w - 10
        sums = torch.zeros(n, dtype=torch.float64, device=p.device)
        maps = torch.empty(3, n, c, oh, ow, device=p.device, dtype=torch.float32)
        ops._ck(ops.lib().muvo_ssim_maps(ops._f(p), ops._f(t), ops._f(win), ops._p(sums), ops._f(maps[0]), ops._f(maps[1]), ops._f(maps[2]),
                                        n, c, h, w, ops._fl(c1), ops._fl(c2), ops._st()))
        ctx.save_for_backward(p, t, win, maps)
        return (sums / (c * oh * ow)).mean().float()

    @staticmethod
    def backward(ctx, g):
        p, t, win, maps = ctx.saved_tensors
        n, c, h, w = p.shape
        dp = torch.empty_like(p)
        scale = float(g) / (n * c * (h - 10) * (w - 10))
        ops._ck(ops.lib().muvo_ssim_bwd(ops._f(p), ops._f(t), ops._f(win), ops._f(maps[0]), ops._f(maps[1]), ops._f(maps[2]), ops._f(dp), n, c,
                                       h, w, ops._fl(scale), ops._st()))
        return dp, None, None, None, None


class SSIMLoss(nn.Module):
    """muvo/losses.py:292-348 (returns the mean SSIM; the trainer uses 1 - it, trainer.py:312-318)."""

    def __init__(self, channel=1, window_size=11, sigma=1.5, L=1, non_negative=False):
        super().__init__()
        if window_size != 11 or non_negative:
            raise NotImplementedError('the kernels implement the reference configuration (window 11, non_negative=False)')
        from .metrics import _gauss_window
        self.channel, self.C1, self.C2 = channel, (0.01 * L) ** 2, (0.03 * L) ** 2
        self.register_buffer('window2d', _gauss_window(window_size, sigma), persistent=False)

    def forward(self, prediction, target):
        b, s, c, h, w = prediction.shape
        win = self.window2d.to(prediction.device)
        return _SSIMMeanFn.apply(prediction.reshape(b * s, c, h, w).float(), target.reshape(b * s, c, h, w), win, self.C1, self.C2)
