"""Model building blocks of the hot path, mirroring muvo/models/common.py (names = state_dict keys) with every
op executed by the gfx950 kernels in muvo_amd.ops.

Covered (base_1d): RouteEncode (common.py:12-23), Policy (:53-68), DecoderDS (:102-130), DecoderBlock3d (:161-172),
ConvInstanceNorm3d (:190-202), AdaptiveInstanceNorm3d (:227-246), RGBHead/LidarReHead/VoxelSemHead (:274-303,354-367),
VoxelDecoder1 (:498-546), ConvDecoder (:549-632), PositionEmbeddingSine (:636-678).
"""
import math

import torch
import torch.nn as nn

from muvo_amd import nn as hnn
from muvo_amd import ops
from muvo_amd.models.resnet import ResNet18Features


class RouteEncode(nn.Module):
    def __init__(self, out_channels, backbone='resnet18'):
        super().__init__()
        assert backbone == 'resnet18'
        self.backbone = ResNet18Features(in_chans=3, out_indices=(4,))
        self.out_channels = out_channels
        self.fc = hnn.Linear(512, out_channels)

    def forward(self, route):
        x = self.backbone(route)[0]
        return self.fc(ops.global_avg_pool(x))


class Policy(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        c = in_channels
        self.fc = nn.Sequential(hnn.Linear(c, c), hnn.Placeholder(), hnn.Linear(c, c), hnn.Placeholder(),
                                hnn.Linear(c, c // 2), hnn.Placeholder(), hnn.Linear(c // 2, 2), hnn.Placeholder())

    def forward(self, x):
        x = self.fc[0](x, act=ops.ACT_RELU)
        x = self.fc[2](x, act=ops.ACT_RELU)
        x = self.fc[4](x, act=ops.ACT_RELU)
        return self.fc[6](x, act=ops.ACT_TANH)


def _conv_bn_relu(cin, cout):
    return nn.Sequential(hnn.Conv2d(cin, cout, 3, 1, 1, bias=False), hnn.BatchNorm2d(cout), hnn.Placeholder())


class DecoderDS(nn.Module):
    def __init__(self, feature_info, out_channels):
        super().__init__()
        self.conv1 = _conv_bn_relu(feature_info[0]['num_chs'], out_channels)
        self.downsample_skip_convs = nn.ModuleList(
            _conv_bn_relu(feature_info[i]['num_chs'], out_channels) for i in range(1, len(feature_info)))
        self.out_channels = out_channels

    def feat_consumers(self):
        """the convolution that reads each input feature map (ResNet18Features.forward(feat_consumers=...))"""
        return [(self.conv1[0],)] + [(c[0],) for c in self.downsample_skip_convs]

    def forward(self, xs):
        x = self.conv1[1](self.conv1[0](xs[0]), relu=True, from_conv=True)
        for i, conv in enumerate(self.downsample_skip_convs):
            stride = xs[i].shape[-1] // xs[i + 1].shape[-1]
            pooled = ops.max_pool2d(x, stride)
            # relu(bn(conv(x_{i+1}))) + max_pool(x): residual added AFTER the ReLU (common.py:128)
            x = conv[1](conv[0](xs[i + 1]), residual=pooled, res_mode=2, relu=True, from_conv=True)
        return x


class Decoder(nn.Module):
    """common.py:71-99: coarse-to-fine skip decoder with bilinear upsampling (BEV lifting variant)."""

    def __init__(self, feature_info, out_channels):
        super().__init__()
        n_upsample_skip_convs = len(feature_info) - 1
        self.conv1 = _conv_bn_relu(feature_info[-1]['num_chs'], out_channels)
        self.upsample_skip_convs = nn.ModuleList(
            _conv_bn_relu(feature_info[-i]['num_chs'], out_channels) for i in range(2, n_upsample_skip_convs + 2))
        self.out_channels = out_channels

    def forward(self, xs):
        x = self.conv1[1](self.conv1[0](xs[-1]), relu=True)
        for i, conv in enumerate(self.upsample_skip_convs):
            up = ops.interpolate_bilinear(x, xs[-(i + 2)].shape[-2:])
            # relu(bn(conv(skip))) + upsample(x): residual added AFTER the ReLU (common.py:96)
            x = conv[1](conv[0](xs[-(i + 2)]), residual=up, res_mode=2, relu=True)
        return x


class StyleBank:
    """The (scale, bias) styles of ALL AdaptiveInstanceNorm layers of one decoder, computed from the latent in one grouped
    launch (ops.grouped_linear: every layer's `latent_affine` reads the same latent, common.py:205-246) instead of one tiny
    GEMM per layer and pass.  Stands in for the latent `w` on its way through the decoder: the norm layers ask it for their
    style; anything else still finds the latent in `.latent`."""

    def __init__(self, latent, modules, styles):
        self.latent = latent
        self.shape = latent.shape
        self._styles = {id(m): s for m, s in zip(modules, styles)}

    def style_of(self, norm):
        return self._styles[id(norm)]

    @staticmethod
    def build(decoder, w):
        """w itself when the grouped kernels do not apply (then every layer runs its own Linear, as before)"""
        if isinstance(w, StyleBank):
            return w
        norms = decoder.__dict__.get('_adain_layers')
        if norms is None:
            norms = [m for m in decoder.modules() if isinstance(m, (AdaptiveInstanceNorm3d, AdaptiveInstanceNorm))]
            decoder.__dict__['_adain_layers'] = norms          # plain attribute: not a registered submodule list
        linears = [m.latent_affine for m in norms]
        if not norms or not ops.grouped_linear_supported(w, linears):
            return w
        return StyleBank(w, norms, ops.grouped_linear(w, linears))


def _style(norm, style):
    return style.style_of(norm) if isinstance(style, StyleBank) else norm.latent_affine(style)


class AdaptiveInstanceNorm3d(nn.Module):
    def __init__(self, latent_n_channels, out_channels, epsilon=1e-8):
        super().__init__()
        self.out_channels = out_channels
        self.epsilon = epsilon
        self.latent_affine = hnn.Linear(latent_n_channels, 2 * out_channels)

    def forward(self, x, style, pre_act=ops.ACT_NONE, pre_slope=0.0, moments=None):
        return ops.adain(x, _style(self, style), self.epsilon, style.shape[0], pre_act, pre_slope, moments)


class ConvInstanceNorm3d(nn.Module):
    def __init__(self, in_channels, out_channels, latent_n_channels):
        super().__init__()
        self.conv_act = nn.Sequential(hnn.Conv3d(in_channels, out_channels, 3, 1, 1), hnn.Placeholder())
        self.adaptive_norm = AdaptiveInstanceNorm3d(latent_n_channels, out_channels)

    def forward(self, x, w, lazy=None, next_conv=None):
        """lazy: (raw, aff) when x is the placeholder of the previous layer's lazy AdaIN (this convolution applies it).
        next_conv: the convolution that is the ONLY consumer of this layer's output; when it can apply an AdaIN while staging
        (ops.conv_affine_supported) the result is (placeholder, (raw, aff)) and the normalised tensor is never written."""
        # LeakyReLU is fused into the conv epilogue; its derivative is chained inside the AdaIN backward kernel
        # (top two levels, bf16x3 voxel kernels: the conv epilogue also delivers the instance-norm statistics of its output)
        moments = ops.conv_moments_buffer(x, self.conv_act[0].geom)
        x = self.conv_act[0](x, act=ops.ACT_LEAKY, slope=0.2, act_bwd_fused=True, moments=moments, lazy=lazy)
        if next_conv is not None and ops.conv_affine_supported(x, next_conv.geom, moments):
            an = self.adaptive_norm
            y, raw, aff = ops.adain_lazy(x, _style(an, w), an.epsilon, ops.ACT_LEAKY, 0.2, moments)
            return y, (raw, aff)
        y = self.adaptive_norm(x, w, pre_act=ops.ACT_LEAKY, pre_slope=0.2, moments=moments)
        return y if next_conv is None else (y, None)

    def forward_with_head(self, x, w, head_conv, lazy=None):
        """This layer followed by a 1x1x1 head that is the ONLY consumer of its output (VoxelDecoder1's last stage): returns
        the head's logits; the normalised tensor is never materialised when the fused kernels apply (ops.AdaINHeadFn)."""
        moments = ops.conv_moments_buffer(x, self.conv_act[0].geom)
        x = self.conv_act[0](x, act=ops.ACT_LEAKY, slope=0.2, act_bwd_fused=True, moments=moments, lazy=lazy)
        if ops.adain_head_supported(x, head_conv.weight, moments):
            an = self.adaptive_norm
            return ops.adain_head(x, _style(an, w), head_conv.weight, head_conv.bias, an.epsilon, moments, ops.ACT_LEAKY, 0.2)
        return head_conv(self.adaptive_norm(x, w, pre_act=ops.ACT_LEAKY, pre_slope=0.2, moments=moments))


class AdaptiveInstanceNorm(nn.Module):
    """common.py:205-224 (2-D): the 3-D kernel with a unit depth."""

    def __init__(self, latent_n_channels, out_channels, epsilon=1e-8):
        super().__init__()
        self.out_channels = out_channels
        self.epsilon = epsilon
        self.latent_affine = hnn.Linear(latent_n_channels, 2 * out_channels)

    def forward(self, x, style, pre_act=ops.ACT_NONE, pre_slope=0.0):
        x5 = x.unsqueeze(-3)                       # (N, C, 1, H, W) or the broadcast parameter (C, 1, H, W)
        y = ops.adain(x5, _style(self, style), self.epsilon, style.shape[0], pre_act, pre_slope)
        return y.squeeze(2)


class ConvInstanceNorm(nn.Module):
    """common.py:175-187."""

    def __init__(self, in_channels, out_channels, latent_n_channels):
        super().__init__()
        self.conv_act = nn.Sequential(hnn.Conv2d(in_channels, out_channels, 3, 1, 1), hnn.Placeholder())
        self.adaptive_norm = AdaptiveInstanceNorm(latent_n_channels, out_channels)

    def forward(self, x, w):
        x = self.conv_act[0](x, act=ops.ACT_LEAKY, slope=0.2, act_bwd_fused=True)
        return self.adaptive_norm(x, w, pre_act=ops.ACT_LEAKY, pre_slope=0.2)


class DecoderBlock(nn.Module):
    """common.py:147-159: bilinear x2 upsample + two ConvInstanceNorm."""

    def __init__(self, in_channels, out_channels, latent_n_channels, upsample=False):
        super().__init__()
        self.upsample = upsample
        self.conv1 = ConvInstanceNorm(in_channels, out_channels, latent_n_channels)
        self.conv2 = ConvInstanceNorm(out_channels, out_channels, latent_n_channels)

    def forward(self, x, w):
        if self.upsample:
            x = ops.interpolate_bilinear(x, (2 * x.shape[-2], 2 * x.shape[-1]))
        return self.conv2(self.conv1(x, w), w)


class SegmentationHead(nn.Module):
    """common.py:249-271: semantic logits, instance offset and (sigmoid) instance centre."""

    def __init__(self, in_channels, n_classes, downsample_factor):
        super().__init__()
        self.downsample_factor = downsample_factor
        self.segmentation_head = nn.Sequential(hnn.Conv2d(in_channels, n_classes, 1, 1, 0))
        self.instance_offset_head = nn.Sequential(hnn.Conv2d(in_channels, 2, 1, 1, 0))
        self.instance_center_head = nn.Sequential(hnn.Conv2d(in_channels, 1, 1, 1, 0), hnn.Placeholder())

    def forward(self, x):
        f = self.downsample_factor
        return {f'bev_segmentation_{f}': self.segmentation_head[0](x),
                f'bev_instance_offset_{f}': self.instance_offset_head[0](x),
                f'bev_instance_center_{f}': self.instance_center_head[0](x, act=ops.ACT_SIGMOID)}


class BevDecoder(nn.Module):
    """common.py:370-424 (head='bev')."""

    def __init__(self, latent_n_channels, semantic_n_channels, constant_size=(3, 3), head='bev'):
        super().__init__()
        assert head == 'bev'
        n = 512
        self.constant_tensor = nn.Parameter(torch.randn((n, *constant_size), dtype=torch.float32))
        self.first_norm = AdaptiveInstanceNorm(latent_n_channels, out_channels=n)
        self.first_conv = ConvInstanceNorm(n, n, latent_n_channels)
        self.middle_conv = nn.ModuleList(DecoderBlock(n, n, latent_n_channels, upsample=True) for _ in range(3))
        self.conv1 = DecoderBlock(n, 256, latent_n_channels, upsample=True)
        self.head_4 = SegmentationHead(256, semantic_n_channels, downsample_factor=4)
        self.conv2 = DecoderBlock(256, 128, latent_n_channels, upsample=True)
        self.head_2 = SegmentationHead(128, semantic_n_channels, downsample_factor=2)
        self.conv3 = DecoderBlock(128, 64, latent_n_channels, upsample=True)
        self.head_1 = SegmentationHead(64, semantic_n_channels, downsample_factor=1)

    def forward(self, w):
        w = StyleBank.build(self, w)
        x = self.first_norm(self.constant_tensor, w)      # the parameter is broadcast over the batch inside the kernel
        x = self.first_conv(x, w)
        for module in self.middle_conv:
            x = module(x, w)
        x = self.conv1(x, w)
        output_4 = self.head_4(x)
        x = self.conv2(x, w)
        output_2 = self.head_2(x)
        x = self.conv3(x, w)
        output_1 = self.head_1(x)
        return {**output_4, **output_2, **output_1}


class DecoderBlock3d(nn.Module):
    def __init__(self, in_channels, out_channels, latent_n_channels, upsample=False):
        super().__init__()
        self.upsample = upsample
        self.conv1 = ConvInstanceNorm3d(in_channels, out_channels, latent_n_channels)
        self.conv2 = ConvInstanceNorm3d(out_channels, out_channels, latent_n_channels)

    def forward(self, x, w, head_conv=None):
        if self.upsample:
            x = ops.upsample3d_x2(x)
        # conv1 -> AdaIN -> conv2: conv2 is the only consumer of the first AdaIN and applies it while staging when it can
        y, lazy = self.conv1(x, w, next_conv=self.conv2.conv_act[0])
        if head_conv is not None:
            return self.conv2.forward_with_head(y, w, head_conv, lazy=lazy)
        return self.conv2(y, w, lazy=lazy)


class _Head(nn.Module):
    """1x1(x1) conv head; attr/key names follow RGBHead / LidarReHead / VoxelSemHead."""

    def __init__(self, attr, key, conv, downsample_factor):
        super().__init__()
        self.downsample_factor = downsample_factor
        self._attr, self._key = attr, key
        setattr(self, attr, nn.Sequential(conv))

    def forward(self, x):
        return {f'{self._key}_{self.downsample_factor}': getattr(self, self._attr)[0](x)}

    def branch(self, x):
        """(x, outputs) for a head that reads a feature map the trunk goes on using: its backward accumulates the head's
        data gradient into the trunk's (ops.HeadBranchFn)."""
        m = getattr(self, self._attr)[0]
        x, y = ops.head_branch(x, m.weight, m.bias, m.geom, m._packed)
        return x, {f'{self._key}_{self.downsample_factor}': y}


def RGBHead(in_channels, n_classes, downsample_factor):
    return _Head('rgb_head', 'rgb', hnn.Conv2d(in_channels, n_classes, 1, 1, 0), downsample_factor)


def LidarReHead(in_channels, n_classes, downsample_factor):
    return _Head('lidar_re_head', 'lidar_reconstruction', hnn.Conv2d(in_channels, n_classes, 1, 1, 0), downsample_factor)


def LidarSegHead(in_channels, n_classes, downsample_factor):      # common.py:306-319
    return _Head('seg_head', 'lidar_segmentation', hnn.Conv2d(in_channels, n_classes, 1, 1, 0), downsample_factor)


def SemHead(in_channels, n_classes, downsample_factor):           # common.py:322-335
    return _Head('sem_head', 'semantic_image', hnn.Conv2d(in_channels, n_classes, 1, 1, 0), downsample_factor)


def DepthHead(in_channels, n_classes, downsample_factor):         # common.py:338-351
    return _Head('depth_head', 'depth', hnn.Conv2d(in_channels, n_classes, 1, 1, 0), downsample_factor)


def VoxelSemHead(in_channels, n_classes, downsample_factor):
    return _Head('segmentation_head', 'voxel', hnn.Conv3d(in_channels, n_classes, 1, 1, 0), downsample_factor)


class VoxelDecoder1(nn.Module):
    def __init__(self, latent_n_channels, semantic_n_channels, feature_channels=512, constant_size=(3, 3, 1)):
        super().__init__()
        n = feature_channels
        self.constant_tensor = nn.Parameter(torch.randn((2 * n, *constant_size), dtype=torch.float32))
        self.first_norm = AdaptiveInstanceNorm3d(latent_n_channels, out_channels=2 * n)
        self.first_conv = ConvInstanceNorm3d(2 * n, n, latent_n_channels)
        self.middle_conv = nn.ModuleList(DecoderBlock3d(n, n, latent_n_channels, upsample=True) for _ in range(3))
        self.conv1 = DecoderBlock3d(n, n // 2, latent_n_channels, upsample=True)
        self.head_4 = VoxelSemHead(n // 2, semantic_n_channels, downsample_factor=4)
        self.conv2 = DecoderBlock3d(n // 2, n // 4, latent_n_channels, upsample=True)
        self.head_2 = VoxelSemHead(n // 4, semantic_n_channels, downsample_factor=2)
        self.conv3 = DecoderBlock3d(n // 4, n // 8, latent_n_channels, upsample=True)
        self.head_1 = VoxelSemHead(n // 8, semantic_n_channels, downsample_factor=1)

    def forward(self, w):
        # constant_tensor is broadcast over the batch inside the AdaIN kernel (no repeat() copy)
        w = StyleBank.build(self, w)
        x = self.first_norm(self.constant_tensor, w)
        x = self.first_conv(x, w)
        for module in self.middle_conv:
            x = module(x, w)
        x = self.conv1(x, w)
        x, output_4 = self.head_4.branch(x)
        x = self.conv2(x, w)
        x, output_2 = self.head_2.branch(x)
        # last stage: the block's output feeds the head only -> AdaIN + head fused (the 1.5 GB normalised tensor is not written)
        h1 = self.head_1
        output_1 = {f'{h1._key}_{h1.downsample_factor}': self.conv3(x, w, head_conv=getattr(h1, h1._attr)[0])}
        return {**output_4, **output_2, **output_1}


class _Seed1x1ConvTFn(torch.autograd.Function):
    """ConvTranspose2d(C, Co, k) applied to a (N, C, 1, 1) input == one GEMM
    y[n][(co,t)] = sum_ci x[n][ci] W[ci][(co,t)] + b[co], fused ELU (common.py:578-581)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = x.contiguous()
        n, ci = x.shape[0], x.shape[1]
        co, kh, kw = weight.shape[1:]
        ncol = co * kh * kw
        y = torch.empty(n, co, kh, kw, device=x.device, dtype=torch.float32)
        ops.gemm(x, weight, y, n, ncol, ci, ci, 1, ncol, 1, ncol, bias=bias, bias_div=kh * kw, act=act)
        ctx.weight, ctx.bias, ctx.act = weight, bias, act
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes as C
        x, y = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        n, ci = x.shape[0], x.shape[1]
        co, kh, kw = weight.shape[1:]
        ncol = co * kh * kw
        dy = dy.contiguous()
        dz = torch.empty_like(dy)
        L = ops.lib()
        ops._ck(L.muvo_act_bwd(ops._f(y), ops._f(dy), ops._f(dz), ops._i64(dy.numel()), ctx.act, ops._fl(0.0), ops._st()))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ops.gemm(dz, weight, dx, n, ci, ncol, ncol, 1, 1, ncol, ci)
        ops.gemm(x, dz, ops.grad_of(weight), ci, ncol, n, 1, ci, ncol, 1, ncol, mode=1)
        if bias is not None:
            ops._ck(L.muvo_bias_grad_nchw(ops._f(dz), ops._f(ops.grad_of(bias)), n, co, ops._i64(kh * kw), ops._st()))
        return dx, None, None, None


class ConvDecoder(nn.Module):
    def __init__(self, latent_n_channels, out_channels, constant_size=(5, 13), head='rgb'):
        super().__init__()
        n_channels = 512
        self.linear = nn.Sequential(hnn.Linear(latent_n_channels, n_channels), hnn.Placeholder())
        P = hnn.Placeholder
        self.pre_transpose_conv = nn.Sequential(
            hnn.ConvTranspose2d(n_channels, n_channels, constant_size), P(),
            hnn.ConvTranspose2d(n_channels, n_channels, 5, 2, 2, 1), P(),
            hnn.ConvTranspose2d(n_channels, n_channels, 5, 2, 2, 1), P(),
            hnn.ConvTranspose2d(n_channels, n_channels, 6, 2, 2), P())
        head_module = {'rgb': RGBHead, 'lidar_re': LidarReHead, 'lidar_seg': LidarSegHead, 'sem_image': SemHead,
                       'depth': DepthHead}[head]      # common.py:588-594
        self.trans_conv1 = nn.Sequential(hnn.ConvTranspose2d(n_channels, 256, 6, 2, 2), P())
        self.head_4 = head_module(256, out_channels, downsample_factor=4)
        self.trans_conv2 = nn.Sequential(hnn.ConvTranspose2d(256, 128, 6, 2, 2), P())
        self.head_2 = head_module(128, out_channels, downsample_factor=2)
        self.trans_conv3 = nn.Sequential(hnn.ConvTranspose2d(128, 64, 6, 2, 2), P())
        self.head_1 = head_module(64, out_channels, downsample_factor=1)

    def forward(self, x):
        x = self.linear[0](x)  # (N, 512); Unflatten to (N,512,1,1) is a view
        seed = self.pre_transpose_conv[0]
        x = _Seed1x1ConvTFn.apply(x.view(x.shape[0], -1, 1, 1), seed.weight, seed.bias, ops.ACT_ELU)
        for i in (2, 4, 6):
            x = self.pre_transpose_conv[i](x, act=ops.ACT_ELU)
        x, output_4 = self._stage(self.trans_conv1[0], self.head_4, x)
        x, output_2 = self._stage(self.trans_conv2[0], self.head_2, x)
        _, output_1 = self._stage(self.trans_conv3[0], self.head_1, x)
        return {**output_4, **output_2, **output_1}

    @staticmethod
    def _stage(conv, head, x):
        """ELU(ConvTranspose(x)) and the 1x1 head on it.  Fused form (ops.ConvHeadFn): the head's data gradient is formed inside
        the stage's backward split pass instead of a pass over the whole feature map; otherwise the head hangs off the trunk
        through HeadBranchFn as before."""
        hc = getattr(head, head._attr)[0]
        if ops.conv_head_supported(x, conv.geom, hc.geom):
            y, logits = ops.conv_head(x, conv.weight, conv.bias, conv.geom, conv._packed, ops.ACT_ELU, 0.0,
                                      hc.weight, hc.bias, hc.geom, hc._packed)
            return y, {f'{head._key}_{head.downsample_factor}': logits}
        return head.branch(conv(x, act=ops.ACT_ELU))


def position_embedding_sine(h, w, num_pos_feats, temperature=10000, scale=2 * math.pi):
    """Constant (2*num_pos_feats, h*w) table of PositionEmbeddingSine(normalize=True) (common.py:636-678),
    evaluated once on the host in float32 with the reference's operation order."""
    ones = torch.ones((1, h, w), dtype=torch.float32)
    y_embed = ones.cumsum(1, dtype=torch.float32)
    x_embed = ones.cumsum(2, dtype=torch.float32)
    eps = 1e-6
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
    pos = torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)
    return pos.reshape(2 * num_pos_feats, h * w).contiguous()
