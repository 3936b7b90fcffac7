"""BEV lifting of image features on the GPU: FrustumPooling (muvo/models/frustum_pooling.py:67-217) as
Mile.encode uses it (mile.py:506-522), on the HIP kernels of csrc/bev.hip through the C ABI.  Same constructor and buffers
as the reference module.  The call site's outer product `depth.unsqueeze(1) * x.unsqueeze(2)` (mile.py:519) is fused into
the pooling kernel, so the entry point takes the feature map and the depth distribution separately (`lift`)."""
import numpy as np
import torch
import torch.nn as nn

from . import ops


def bev_params_to_intrinsics(size, scale, offsetx):
    """geometry_utils.py:8-19."""
    return np.array([[1 / scale, 0, size[0] / 2 + offsetx], [0, -1 / scale, size[1] / 2], [0, 0, 1]], dtype=np.float32)


def gen_dx_bx(size, scale, offsetx):
    """frustum_pooling.py:10-21."""
    xbound = [-size[0] * scale / 2 - offsetx * scale, size[0] * scale / 2 - offsetx * scale, scale]
    ybound = [-size[1] * scale / 2, size[1] * scale / 2, scale]
    zbound = [-10.0, 10.0, 20.0]
    dx = torch.Tensor([row[2] for row in [xbound, ybound, zbound]])
    bx = torch.Tensor([row[0] + row[2] / 2.0 for row in [xbound, ybound, zbound]])
    nx = torch.LongTensor([np.round((row[1] - row[0]) / row[2]) for row in [xbound, ybound, zbound]])
    return dx, bx, nx


class _LiftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, depth, mask, cells, ncell):
        B, C, H, W = feat.shape
        D = depth.shape[1]
        feat, depth = feat.contiguous(), depth.contiguous()
        acc = torch.empty(B, ncell, C, device=feat.device, dtype=torch.float32)
        out = torch.empty(B, C, ncell, device=feat.device, dtype=torch.float32)
        ops._ck(ops.lib().muvo_frustum_pool_fwd(ops._f(feat), ops._f(depth), ops._p(mask), ops._p(cells), ops._f(acc), ops._f(out), B, C, D,
                                               ops._i64(H * W), ncell, ops._st()))
        ctx.save_for_backward(feat, depth, mask, cells)
        ctx.ncell = ncell
        return out

    @staticmethod
    def backward(ctx, gout):
        feat, depth, mask, cells = ctx.saved_tensors
        B, C, H, W = feat.shape
        D = depth.shape[1]
        g_cl = torch.empty(B, ctx.ncell, C, device=feat.device, dtype=torch.float32)
        dfeat, ddepth = torch.empty_like(feat), torch.empty_like(depth)
        ops._ck(ops.lib().muvo_frustum_pool_bwd(ops._f(feat), ops._f(depth), ops._p(mask), ops._p(cells), ops._f(gout.contiguous()),
                                               ops._f(g_cl), ops._f(dfeat), ops._f(ddepth), B, C, D, ops._i64(H * W), ctx.ncell, ops._st()))
        return dfeat, ddepth, None, None, None


class FrustumPooling(nn.Module):
    """frustum_pooling.py:67-217 (one camera per frame, as in mile.py:520-522)."""

    def __init__(self, size, scale, offsetx, dbound, downsample, use_quickcumsum=True):
        super().__init__()
        self.register_buffer('bev_intrinsics', torch.tensor(bev_params_to_intrinsics(size, scale, offsetx)))
        dx, bx, nx = gen_dx_bx(size, scale, offsetx)
        self.nx_constant = nx.numpy().tolist()
        self.register_buffer('dx', dx, persistent=False)
        self.register_buffer('bx', bx, persistent=False)
        self.register_buffer('nx', nx, persistent=False)
        self.use_quickcumsum = use_quickcumsum
        self.dbound = dbound
        ds = torch.arange(self.dbound[0], self.dbound[1], self.dbound[2], dtype=torch.float32)
        self.D = len(ds)
        self.register_buffer('ds', ds, persistent=False)
        self.downsample = downsample
        self._grid = None

    def _frustum_axes(self, fH, fW, device):
        """initialize_frustum (:92-106): pixel coordinates of the feature-map grid in the full-resolution image (computed
        on the host exactly as the reference's CPU linspace does)."""
        if self._grid is None or self._grid[0] != (fH, fW):
            xs = torch.linspace(0, fW * self.downsample - 1, fW, dtype=torch.float).to(device)
            ys = torch.linspace(0, fH * self.downsample - 1, fH, dtype=torch.float).to(device)
            self._grid = ((fH, fW), xs, ys)
        return self._grid[1], self._grid[2]

    def cells(self, intrinsics, pose, fH, fW):
        """BEV cell of every frustum point: int32 (B, D, fH, fW), -1 outside (get_geometry :108-128 + voxel_pooling :139-158)."""
        B = intrinsics.shape[0]
        dev = intrinsics.device
        fx, fy, cx, cy = intrinsics[:, 0, 0], intrinsics[:, 1, 1], intrinsics[:, 0, 2], intrinsics[:, 1, 2]
        one, zero = torch.ones_like(fx), torch.zeros_like(fx)
        kinv = torch.stack((torch.stack((1 / fx, zero, -cx / fx), -1), torch.stack((zero, 1 / fy, -cy / fy), -1),
                            torch.stack((zero, zero, one), -1)), -2)                     # geometry_utils.py:22-34
        combine = pose[:, :3, :3].matmul(kinv).contiguous().float()
        trans = pose[:, :3, 3].contiguous().float()
        xs, ys = self._frustum_axes(fH, fW, dev)
        cells = torch.empty(B, self.D, fH, fW, dtype=torch.int32, device=dev)
        bi = self.bev_intrinsics
        nx = self.nx_constant
        ops._ck(ops.lib().muvo_frustum_cells(ops._f(combine), ops._f(trans), ops._f(xs), ops._f(ys), ops._f(self.ds.to(dev)), ops._p(cells),
                                            B, self.D, fH, fW, ops._fl(bi[0, 0]), ops._fl(bi[0, 2]), ops._fl(bi[1, 1]), ops._fl(bi[1, 2]),
                                            ops._fl(self.bx[2]), ops._fl(self.dx[2]), nx[0], nx[1], nx[2], ops._st()))
        return cells

    def lift(self, feat, depth, intrinsics, pose, mask=None):
        """feat (B, C, H, W), depth (B, D, H, W) distribution, intrinsics (B, 3, 3), pose (B, 4, 4), mask (B, D, H, W) bool or
        None -> BEV features (B, C * nz, ny, nx)  (mile.py:506-522 + FrustumPooling.forward)."""
        B, C, H, W = feat.shape
        assert depth.shape == (B, self.D, H, W)
        cells = self.cells(intrinsics, pose, H, W)
        nx = self.nx_constant
        ncell = nx[0] * nx[1] * nx[2]
        m = None
        if mask is not None and mask.numel():
            m = mask.to(torch.uint8).contiguous()
        out = _LiftFn.apply(feat.float(), depth.float(), m, cells, ncell).view(B, C, nx[2], nx[1], nx[0])
        return torch.cat(out.unbind(dim=2), 1) if nx[2] > 1 else out.view(B, C, nx[1], nx[0])

    def get_depth_map(self, depth):
        """:211-217: expected depth, bilinear x downsample."""
        B, D, H, W = depth.shape
        e = torch.empty(B, 1, H, W, device=depth.device, dtype=torch.float32)
        ops._ck(ops.lib().muvo_depth_expectation(ops._f(depth.float().contiguous()), ops._f(self.ds.to(depth.device)), ops._f(e), B, D,
                                                ops._i64(H * W), ops._st()))
        return ops.resize_bilinear(e, H * self.downsample, W * self.downsample)
