"""dev (CPU, plain PyTorch): the identity the next voxel-decoder step rests on (DESIGN.md section 7 "Next", item 1).

    conv3d_3x3x3(trilinear_x2(x), w)  ==  8 interleaved 3x3x3 convolutions of the LOW-resolution tensor x

Per axis the trilinear x 2 interpolation (align_corners=False) is out[2i] = 0.25 x[i-1] + 0.75 x[i], out[2i+1] = 0.75 x[i] + 0.25 x[i+1]
with the index clamped at the borders = edge replication of x.  A 3-tap convolution of `out` at output 2i + p (p = 0, 1) therefore
reads x[i-1], x[i], x[i+1] through a 3 x 3 matrix A_p (rows: low-res offset -1..1, columns: high-res tap -1..1); in 3-D the composite
weights are W_eff[p][co][ci][a,b,c] = sum_{t,u,v} A_px[a,t] A_py[b,u] A_pz[c,v] w[co][ci][t,u,v].  The zero padding of the convolution
lives in HIGH-resolution space (out[-1] = out[2N] = 0, not interpolated), so the outermost high-res shell differs and is excluded here
(it needs its own small kernel).  Prints the largest deviation inside and the share of voxels in the shell."""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
N, CI, CO, X, Y, Z = 2, 4, 3, 6, 5, 4
x = torch.randn(N, CI, X, Y, Z, dtype=torch.float64)
w = torch.randn(CO, CI, 3, 3, 3, dtype=torch.float64)

ref = F.conv3d(F.interpolate(x, scale_factor=2.0, mode='trilinear', align_corners=False), w, padding=1)

# A[p][a, t]: weight of low-res offset a - 1 in the high-res value at offset t - 1 from output 2 i + p
A = torch.zeros(2, 3, 3, dtype=torch.float64)
for p in (0, 1):
    for t in (-1, 0, 1):
        h = p + t                               # high-res position relative to 2 i
        i0, par = divmod(h, 2)                  # h = 2 i0 + par
        if par == 0:                            # even position 2 i0: 0.25 x[i0 - 1] + 0.75 x[i0]
            A[p, i0 - 1 + 1, t + 1] += 0.25
            A[p, i0 + 1, t + 1] += 0.75
        else:                                   # odd position 2 i0 + 1: 0.75 x[i0] + 0.25 x[i0 + 1]
            A[p, i0 + 1, t + 1] += 0.75
            A[p, i0 + 1 + 1, t + 1] += 0.25

xr = F.pad(x, (1, 1, 1, 1, 1, 1), mode='replicate')          # index clamping of the interpolation
out = torch.zeros_like(ref)
for px in (0, 1):
    for py in (0, 1):
        for pz in (0, 1):
            weff = torch.einsum('at,bu,cv,oituv->oiabc', A[px], A[py], A[pz], w)
            out[:, :, px::2, py::2, pz::2] = F.conv3d(xr, weff)
inner = (slice(None), slice(None), slice(1, -1), slice(1, -1), slice(1, -1))
print('largest deviation inside the outermost shell: %.2e (values up to %.1f)' % ((out - ref)[inner].abs().max().item(), ref.abs().max().item()))
shell = 1.0 - (2 * X - 2) * (2 * Y - 2) * (2 * Z - 2) / (8.0 * X * Y * Z)
print('shell share at this size: %.1f %%; at 192 x 192 x 64: %.1f %%' % (100 * shell, 100 * (1 - 190 * 190 * 62 / (192 * 192 * 64))))
print('largest deviation in the shell (zero padding in high-res space, to be overwritten by a direct kernel): %.2e' % (out - ref).abs().max().item())
