"""-m gpu: every HIP kernel family against a plain PyTorch fp32 CPU reference of the same op (through the C ABI via
muvo_amd.ops).  Tolerances are stated per test; sizes are small so the CPU side finishes in seconds."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=2e-4, atol=2e-5, name=''):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, f'{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol + rtol * ref * 0.1), f'{name}: max err {err:.3e} (ref max {ref:.3e})'


def test_mfma_selftest(dev):
    from muvo_amd import ops
    assert ops.lib().muvo_selftest_mfma(None) == 0, ops.lib().muvo_last_error()


CONV_CASES = [
    # nd, transposed, cin, cout, k, stride, pad, out_pad, in_sz, bias, act
    (2, False, 3, 64, 7, 2, 3, 0, (38, 50), False, 0),
    (2, False, 4, 64, 7, 2, 3, 0, (16, 64), False, 0),
    (2, False, 64, 64, 3, 1, 1, 0, (19, 25), False, 0),
    (2, False, 64, 128, 3, 2, 1, 0, (20, 26), False, 0),
    (2, False, 64, 128, 1, 2, 0, 0, (20, 26), False, 0),
    (2, False, 48, 160, 3, 1, 1, 0, (10, 26), False, 0),
    # 5x5 stride-2 convs of the BEV down-sampling (mile.py:53-57): data gradient = four sub-pixel phases with 9/6/6/4 taps
    (2, False, 48, 64, 5, 2, 2, 0, (12, 12), True, 1),
    (2, False, 64, 48, 5, 2, 2, 0, (11, 9), True, 0),
    (2, False, 64, 3, 1, 1, 0, 0, (24, 40), True, 0),
    (2, False, 20, 4, 1, 1, 0, 0, (16, 64), True, 0),
    (2, True, 40, 24, 5, 2, 2, 1, (5, 13), True, 3),
    (2, True, 24, 40, 6, 2, 2, 0, (10, 26), True, 3),
    (2, True, 32, 16, 6, 2, 2, 0, (8, 64), True, 3),
    (2, True, 130, 70, 6, 2, 2, 0, (5, 7), True, 0),
    (3, False, 16, 8, 3, 1, 1, 0, (6, 6, 4), True, 2),
    (3, False, 40, 20, 3, 1, 1, 0, (3, 3, 1), True, 2),
    (3, False, 8, 2, 1, 1, 0, 0, (6, 6, 4), True, 0),
    (3, False, 6, 8, 3, 1, 1, 0, (8, 8, 4), True, 2),
    # small-channel Conv3d path (conv_vox.hip: 4x4x1 MFMA fwd/dgrad/wgrad), Z in {64, 32}, ragged X/Y
    (3, False, 16, 8, 3, 1, 1, 0, (5, 13, 64), True, 2),
    (3, False, 8, 8, 3, 1, 1, 0, (4, 7, 32), True, 2),
    (3, False, 32, 16, 3, 1, 1, 0, (3, 9, 32), True, 2),
    (3, False, 16, 16, 3, 1, 1, 0, (6, 5, 64), False, 0),
    (3, False, 8, 8, 3, 1, 1, 0, (1, 1, 64), True, 0),
    (3, False, 16, 8, 3, 1, 1, 0, (19, 12, 32), True, 2),
    # plane-streaming weight gradient only (conv_vox.hip vox_wgrad_ps_only): z lines of 16 voxels (two rows per wave), 32 produced
    # channels as two row blocks; forward / data gradient stay on the implicit-GEMM kernels
    (3, False, 64, 32, 3, 1, 1, 0, (5, 19, 16), True, 2),
    (3, False, 32, 32, 3, 1, 1, 0, (4, 31, 16), True, 2),      # (at (4, 33, 16) one forward value lands on the other side of the LeakyReLU kink)
    (3, False, 16, 8, 3, 1, 1, 0, (3, 16, 16), False, 0),
    (3, False, 32, 32, 3, 1, 1, 0, (3, 7, 32), True, 2),
    # merged sub-pixel phases (ConvTranspose k6 s2 p2 with Cout % 32 == 0: the four phases run as one GEMM)
    (2, True, 48, 64, 6, 2, 2, 0, (7, 9), True, 3),
    (2, True, 64, 32, 6, 2, 2, 0, (5, 13), True, 3),
    # split-K path of the fp32 kernel (tiny pixel grid, long reduction) incl. the finishing bias+activation pass
    (2, False, 256, 64, 3, 1, 1, 0, (6, 7), True, 3),
    (2, True, 160, 96, 6, 2, 2, 0, (3, 4), True, 3),
    # few result pixels, long reduction (ResNet stage 4 at 5 x 13): split-K on both kernel families (bf16x3: bf3_fwd_ksplit)
    (2, False, 512, 512, 3, 1, 1, 0, (5, 13), True, 3),
    (2, True, 384, 256, 5, 2, 2, 1, (4, 6), True, 0),
    # decoder heads (conv_pw.hip: float4 VALU kernels), Cout <= 4, spatial size % 4 == 0 and >= 1024
    (3, False, 8, 2, 1, 1, 0, 0, (16, 16, 8), True, 0),
    (3, False, 32, 2, 1, 1, 0, 0, (12, 12, 8), True, 0),
    (2, False, 64, 3, 1, 1, 0, 0, (40, 52), True, 0),
    (2, False, 130, 4, 1, 1, 0, 0, (32, 64), True, 0),
]


@pytest.fixture(params=['f32', 'bf16x3'])
def conv_mode(request):
    """Both matrix-pipe arithmetics of the conv family: exact fp32 MFMA and the bf16x3 split-product kernel."""
    from muvo_amd import ops
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_BF16X3 if request.param == 'bf16x3' else ops.CONV_F32, min_gflop=0.0)
    yield request.param
    ops.set_conv_mode(old, min_gflop=-1.0)


@pytest.mark.parametrize('case', CONV_CASES, ids=[str(i) for i in range(len(CONV_CASES))])
def test_conv_family(dev, case, conv_mode):
    _check_conv(dev, case, 3)


# Shapes with more eight-wave tiles than CUs (several rounds of 147-KB workgroups per CU).  256x128 tiles (M > 128), 128x256 tiles
# (64 < M <= 128), merged ConvTranspose phases (M = 4 Cout), a channel count that is no multiple of 32 (taps outermost in K),
# ragged last tiles.  (Smooth activations only: with ~10^7 outputs some pre-activations land within the bf16x3 error of the
# kink of ReLU / LeakyReLU, and the derivative mask of those elements then differs from the CPU reference's.)
MANY_TILE_CASES = [
    (2, False, 64, 160, 3, 1, 1, 0, (160, 168), True, 3),     # 630 tiles of 256x128, K = 576
    (2, False, 48, 96, 3, 1, 1, 0, (150, 181), True, 0),      # 319 tiles of 128x256, Cp = 48
    (2, True, 64, 64, 6, 2, 2, 0, (120, 97), True, 3),        # merged phases: M = 256, 273 tiles; dgrad: stride-2 conv
    (2, False, 160, 64, 3, 1, 1, 0, (100, 131), False, 0),    # dgrad runs the 256x128 tiles (M = 160)
    (3, False, 64, 32, 3, 1, 1, 0, (24, 40, 12), True, 3),    # 32 produced channels (half of the 64-row tile), 27 taps
    (2, False, 48, 24, 3, 1, 1, 0, (100, 120), True, 0),      # 24 produced channels, Cp = 48
]


@pytest.mark.parametrize('case', MANY_TILE_CASES, ids=[str(i) for i in range(len(MANY_TILE_CASES))])
def test_conv_many_eight_wave_tiles(dev, case):
    from muvo_amd import ops
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
    try:
        _check_conv(dev, case, 3)
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)


def _check_conv(dev, case, n):
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    nd, transposed, cin, cout, k, stride, pad, out_pad, in_sz, bias, act = case
    torch.manual_seed(0)
    with torch.device(dev):
        if nd == 3:
            m = hnn.Conv3d(cin, cout, k, stride, pad, bias=bias)
        elif transposed:
            m = hnn.ConvTranspose2d(cin, cout, k, stride, pad, out_pad, bias=bias)
        else:
            m = hnn.Conv2d(cin, cout, k, stride, pad, bias=bias)
    x = torch.randn(n, cin, *in_sz)
    xg = x.to(dev).requires_grad_(True)
    slope = 0.2
    y = m(xg, act=act, slope=slope)
    w = m.weight.detach().cpu().requires_grad_(True)
    b = m.bias.detach().cpu().requires_grad_(True) if bias else None
    xc = x.clone().requires_grad_(True)
    if nd == 3:
        yr = F.conv3d(xc, w, b, stride, pad)
    elif transposed:
        yr = F.conv_transpose2d(xc, w, b, stride, pad, out_pad)
    else:
        yr = F.conv2d(xc, w, b, stride, pad)
    if act == 1:
        yr = F.relu(yr)
    elif act == 2:
        yr = F.leaky_relu(yr, slope)
    elif act == 3:
        yr = F.elu(yr)
    _close(y, yr, name='fwd')
    g = torch.randn_like(yr)
    yr.backward(g)
    m.weight.grad = torch.zeros_like(m.weight)
    if bias:
        m.bias.grad = torch.zeros_like(m.bias)
    y.backward(g.to(dev))
    _close(xg.grad, xc.grad, name='dgrad')
    _close(m.weight.grad, w.grad, rtol=5e-4, name='wgrad')
    if bias:
        _close(m.bias.grad, b.grad, rtol=5e-4, name='dbias')
    # second backward accumulates into .grad (kernels add)
    y2 = m(xg, act=act, slope=slope)
    y2.backward(g.to(dev))
    _close(m.weight.grad, 2 * w.grad, rtol=5e-4, name='wgrad accumulate')


GEMM_SHAPES = [(7, 33, 5), (130, 257, 70), (324, 48, 324), (2, 1600, 1088), (300, 1, 16), (64, 64, 1),
               (20, 1536, 1536), (3, 100, 70), (9, 64, 16)]  # skinny (M <= 32) kernels incl. row chunking


@pytest.mark.parametrize('shape', [(3, 64, 64, 20, 26), (2, 64, 64, 64, 160), (2, 32, 128, 48, 176)])
@pytest.mark.parametrize('trunk_used', [True, False])
def test_conv_stage_with_head(dev, trunk_used, shape):
    """A ConvDecoder stage (common.py:608-632): ELU(ConvTranspose2d) whose output feeds a 1x1 head (and, except at the last
    stage, the next stage).  ops.ConvHeadFn forms the head's data gradient inside the stage's backward split pass
    (muvo_conv_prepare_dy_head); checked against the plain PyTorch composition."""
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
    try:
        torch.manual_seed(17)
        nb, cin, cout, hh, ww = shape
        with torch.device(dev):
            conv = hnn.ConvTranspose2d(cin, cout, 6, 2, 2)
            head = hnn.Conv2d(cout, 3, 1, 1, 0)
        x = torch.randn(nb, cin, hh, ww)
        xg = x.to(dev).requires_grad_(True)
        assert ops.conv_head_supported(xg, conv.geom, head.geom)
        if hh >= 48:     # the larger shapes run on the eight-wave tiles: the head's forward rides on the stage's epilogue
            d = conv.geom.plan(nb, (1, hh, ww))[0]
            import ctypes
            assert ops.lib().muvo_conv_forward_head_supported(ctypes.byref(d), 3) == 1
        for p in (conv.weight, conv.bias, head.weight, head.bias):
            p.grad = torch.zeros_like(p)
        y, logits = ops.conv_head(xg, conv.weight, conv.bias, conv.geom, conv._packed, ops.ACT_ELU, 0.0,
                                  head.weight, head.bias, head.geom, head._packed)
        xc = x.clone().requires_grad_(True)
        w, b = conv.weight.detach().cpu().requires_grad_(True), conv.bias.detach().cpu().requires_grad_(True)
        hw, hb = head.weight.detach().cpu().requires_grad_(True), head.bias.detach().cpu().requires_grad_(True)
        yr = F.elu(F.conv_transpose2d(xc, w, b, 2, 2))
        lr = F.conv2d(yr, hw, hb)
        _close(y, yr, name='stage output')
        _close(logits, lr, name='head output')
        gl = torch.randn_like(lr)
        gy = torch.randn_like(yr)
        if trunk_used:
            torch.autograd.backward([yr, lr], [gy, gl])
            torch.autograd.backward([y, logits], [gy.to(dev), gl.to(dev)])
        else:
            lr.backward(gl)
            logits.backward(gl.to(dev))
        _close(xg.grad, xc.grad, rtol=5e-4, name='dx')
        _close(conv.weight.grad, w.grad, rtol=5e-4, name='dW')
        _close(conv.bias.grad, b.grad, rtol=5e-4, atol=1e-3, name='db')
        _close(head.weight.grad, hw.grad, rtol=5e-4, atol=1e-3, name='head dW')
        _close(head.bias.grad, hb.grad, rtol=5e-4, atol=1e-3, name='head db')
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)


def test_head_next_to_a_convolution_of_another_stream(dev):
    """A 1x1 output head must give bit-identical results whether or not a convolution of ANOTHER stream shares the chip with it.
    With packed fp32 VALU instructions in the head kernel about half of such launches lost one product in lanes 48-63
    (tools/dev/coresidency_repro.py, profiles/r03j_lidar_decoder_stream.txt); the library is built without them."""
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
    try:
        torch.manual_seed(0)
        with torch.device(dev):
            vox = hnn.Conv3d(32, 32, 3, 1, 1, bias=True)
            head = hnn.Conv2d(64, 3, 1, 1, 0, bias=True)
        xv3 = torch.randn(2, 32, 96, 96, 32, device=dev)
        xv = torch.randn(20, 64, 160, 800, device=dev)
        side = torch.cuda.Stream(device=dev)
        with torch.no_grad():
            vox(xv3, act=2, slope=0.2)
            ref = head(xv).clone()
            torch.cuda.synchronize()
            bad = 0
            for _ in range(25):
                with torch.cuda.stream(side):
                    for _ in range(3):
                        vox(xv3, act=2, slope=0.2)
                outs = [head(xv) for _ in range(4)]
                torch.cuda.synchronize()
                bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        assert bad == 0, f'{bad} of 100 head launches differ from the idle-GPU result'
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)


def test_reset_accumulators_after_interrupted_step(dev):
    """A step interrupted between a convolution epilogue that fills a layer's moments buffer and the AdaIN that consumes and
    clears it leaves partial sums behind; ops.reset_accumulators() (called by ops._ck on library errors and by the trainer when a
    step raises) restores the all-zero state, so the next forward gives the same result as before."""
    from muvo_amd import ops
    from muvo_amd.models.common import DecoderBlock3d
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
    try:
        torch.manual_seed(3)
        with torch.device(dev):
            blk = DecoderBlock3d(16, 8, 24, upsample=False)
        x, w = torch.randn(2, 16, 24, 24, 32, device=dev), torch.randn(2, 24, device=dev)
        with torch.no_grad():
            ref = blk(x, w).clone()
            real = ops.adain_lazy, ops.adain

            def boom(*a, **k):
                raise MemoryError('injected between the convolution and its AdaIN')
            ops.adain_lazy = ops.adain = boom
            try:
                with pytest.raises(MemoryError):
                    blk(x, w)
            finally:
                ops.adain_lazy, ops.adain = real
            buf = ops.conv_moments_buffer(x, blk.conv1.conv_act[0].geom)
            assert buf is not None and float(buf.abs().sum()) > 0, 'the interrupted forward left no partial sums: test is vacuous'
            ops.reset_accumulators()
            assert float(buf.abs().sum()) == 0
            assert torch.equal(blk(x, w), ref)
        with pytest.raises(RuntimeError):           # a library error takes the same path by itself
            ops._ck(-1)
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)


def test_head_branch_shared_gradient(dev):
    """ops.HeadBranchFn adds the head's data gradient into the trunk's gradient tensor IN PLACE - allowed only when that tensor
    belongs to this consumer alone.  Here the trunk's gradient is shared: `a + c` hands the same tensor object to both inputs,
    so an in-place accumulate would corrupt c's gradient (ops._grad_is_private)."""
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    torch.manual_seed(5)
    with torch.device(dev):
        head = hnn.Conv2d(64, 3, 1, 1, 0)
    for p in (head.weight, head.bias):
        p.grad = torch.zeros_like(p)
    x = torch.randn(2, 64, 32, 64, device=dev)
    seen = []
    real = ops._grad_is_private
    ops._grad_is_private = lambda g: (seen.append(real(g)), seen[-1])[1]
    for shared in (True, False):
        t = x.clone().requires_grad_(True)
        c = torch.randn_like(x).requires_grad_(True)
        a, logits = ops.head_branch(t * 1.0, head.weight, head.bias, head.geom, head._packed)
        assert type(logits.grad_fn).__name__.startswith('HeadBranchFn')
        up = torch.randn_like(x)
        gl = torch.randn_like(logits)
        trunk = (a + c) if shared else (a * 1.0 + c * 1.0)
        torch.autograd.backward([trunk, logits], [up, gl])
        assert torch.equal(c.grad, up), 'the shared gradient tensor was modified in place'
        ref = up + torch.einsum('nkhw,kc->nchw', gl, head.weight.detach().view(3, 64))
        _close(t.grad, ref, rtol=1e-5, atol=1e-5, name='trunk gradient')
    ops._grad_is_private = real
    assert seen == [False, True], seen       # shared -> fresh tensor; a gradient made for this consumer alone -> in place


def test_grouped_linear(dev):
    """All style projections of a decoder in one launch per pass (ops.grouped_linear; common.py:205-246: each
    AdaptiveInstanceNorm applies its own Linear to the same latent): outputs, the summed data gradient and the per-layer
    weight / bias gradients (accumulated into existing .grad) against separate torch Linears."""
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    torch.manual_seed(3)
    m, k = 20, 1536
    outs = [128, 16, 512, 64, 2, 1024, 32, 128]
    with torch.device(dev):
        lins = [hnn.Linear(k, n) for n in outs]
    x = torch.randn(m, k)
    xg = x.to(dev).requires_grad_(True)
    assert ops.grouped_linear_supported(xg, lins)
    for l in lins:
        l.weight.grad, l.bias.grad = torch.full_like(l.weight, 0.25), torch.full_like(l.bias, -0.5)
    ys = ops.grouped_linear(xg, lins)
    xc = x.clone().requires_grad_(True)
    ws = [l.weight.detach().cpu().requires_grad_(True) for l in lins]
    bs = [l.bias.detach().cpu().requires_grad_(True) for l in lins]
    yr = [F.linear(xc, w, b) for w, b in zip(ws, bs)]
    for i, (y, r) in enumerate(zip(ys, yr)):
        _close(y, r, name=f'grouped fwd {i}')
    gs = [torch.randn_like(r) for r in yr]
    torch.autograd.backward(yr, gs)
    torch.autograd.backward(list(ys), [g.to(dev) for g in gs])
    _close(xg.grad, xc.grad, rtol=5e-4, name='grouped dx')
    for i, l in enumerate(lins):
        _close(l.weight.grad - 0.25, ws[i].grad, rtol=5e-4, atol=1e-4, name=f'grouped dW {i}')
        _close(l.bias.grad + 0.5, bs[i].grad, rtol=5e-4, atol=1e-4, name=f'grouped db {i}')
    # a consumer that uses only some of the outputs: the missing gradients count as zero
    ys = ops.grouped_linear(xg.detach().requires_grad_(True), lins[:3])
    ys[1].sum().backward()


@pytest.mark.parametrize('shape', GEMM_SHAPES)
def test_linear(dev, shape):
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    rows, out_f, in_f = shape
    torch.manual_seed(1)
    with torch.device(dev):
        lin = hnn.Linear(in_f, out_f)
    x = torch.randn(rows, in_f)
    for act in (0, 1, 4):
        xg = x.to(dev).requires_grad_(True)
        lin.weight.grad = torch.zeros_like(lin.weight)
        lin.bias.grad = torch.zeros_like(lin.bias)
        y = lin(xg, act=act)
        w = lin.weight.detach().cpu().requires_grad_(True)
        b = lin.bias.detach().cpu().requires_grad_(True)
        xc = x.clone().requires_grad_(True)
        yr = F.linear(xc, w, b)
        yr = F.relu(yr) if act == 1 else (torch.tanh(yr) if act == 4 else yr)
        _close(y, yr, name=f'linear fwd act{act}')
        g = torch.randn_like(yr)
        yr.backward(g)
        y.backward(g.to(dev))
        _close(xg.grad, xc.grad, name='linear dx')
        _close(lin.weight.grad, w.grad, rtol=5e-4, name='linear dW')
        _close(lin.bias.grad, b.grad, rtol=5e-4, name='linear db')


@pytest.mark.parametrize('shape', [(2500, 384, 384), (2101, 2048, 384), (2048, 384, 2048), (3000, 1152, 384), (2200, 64, 72)])
def test_linear_bf16x3_token_matrix(dev, shape):
    """nn.Linear on token matrices with thousands of rows runs on the bf16x3 implicit-GEMM kernels
    (muvo_linear_bf16x3_*): ragged row counts, produced features above / below the 128-workgroup tile switch, a second
    backward accumulating into the same weight gradient."""
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    rows, out_f, in_f = shape
    assert ops.get_conv_mode() == ops.CONV_BF16X3 and rows >= ops.LINEAR_BF16X3_MIN_ROWS
    torch.manual_seed(3)
    with torch.device(dev):
        lin = hnn.Linear(in_f, out_f)
    x = torch.randn(rows, in_f)
    for act in (0, 1):
        xg = x.to(dev).requires_grad_(True)
        lin.weight.grad = torch.zeros_like(lin.weight)
        lin.bias.grad = torch.zeros_like(lin.bias)
        y = lin(xg, act=act)
        assert y.grad_fn.name().startswith('LinearBf16x3Fn')
        w = lin.weight.detach().cpu().double().requires_grad_(True)
        b = lin.bias.detach().cpu().double().requires_grad_(True)
        xc = x.double().requires_grad_(True)
        yr = F.linear(xc, w, b)
        yr = F.relu(yr) if act == 1 else yr
        _close(y, yr, rtol=5e-5, name=f'fwd act{act}')
        g = torch.randn_like(yr)
        if act == 1:
            g = g * (yr.abs() > 1e-3)     # keep the ReLU mask decision away from rounding noise
        yr.backward(g)
        y.backward(g.float().to(dev))
        _close(xg.grad, xc.grad, rtol=5e-5, name='dx')
        _close(lin.weight.grad, w.grad, rtol=1e-4, name='dW')
        _close(lin.bias.grad, b.grad, rtol=1e-4, name='db')
    y2 = lin(x.to(dev))
    y2.backward(torch.ones_like(y2))
    y2 = lin(x.to(dev))
    y2.backward(torch.ones_like(y2))
    ref = x.double().sum(0)[None, :].expand(out_f, in_f)
    _close(lin.weight.grad - w.grad.float().to(dev), 2 * ref, rtol=1e-4, name='dW accumulate')


def test_batched_repack_matches_per_layer_pack(dev):
    """ops.repack_all(): one launch over the device table (muvo_pack_table_*) must leave every packed copy bit-identical to
    what muvo_conv_pack_weights / muvo_linear_bf16x3_pack write per layer - exact-fp32 phases, bf16x3 phases with taps inside
    the channel groups, merged sub-pixel phases of a transposed conv, the strided data-gradient phases, a token Linear."""
    import ctypes as C
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    assert ops.get_conv_mode() == ops.CONV_BF16X3
    torch.manual_seed(5)
    with torch.device(dev):
        layers = [(hnn.Conv2d(3, 64, 7, 2, 3), (2, 3, 64, 96)),              # fp32 family (3 reduction channels)
                  (hnn.Conv2d(64, 64, 3, 1, 1), (2, 64, 40, 52)),            # bf16x3, taps inside 32-channel groups
                  (hnn.Conv2d(64, 128, 3, 2, 1), (2, 64, 40, 52)),           # strided: four data-gradient phases
                  (hnn.ConvTranspose2d(128, 64, 6, 2, 2), (2, 128, 20, 26)),  # merged sub-pixel phases
                  (hnn.Conv2d(512, 512, 3, 1, 1), (2, 512, 2, 4))]           # tiny grid: fp32 family
        lin = hnn.Linear(384, 1152)
        xl = torch.randn(2100, 384, requires_grad=True)
    xs = []
    for m, shp in layers:
        x = torch.randn(*shp, device=dev, requires_grad=True)
        m.weight.grad, m.bias.grad = torch.zeros_like(m.weight), torch.zeros_like(m.bias)
        m(x).sum().backward()                       # first use: per-layer packs, registration
        xs.append(x)
    lin.weight.grad, lin.bias.grad = torch.zeros_like(lin.weight), torch.zeros_like(lin.bias)
    lin(xl).sum().backward()
    with torch.no_grad():
        for m, _ in layers:
            m.weight.mul_(1.5).add_(0.01)
        lin.weight.mul_(0.5).sub_(0.02)
    ops.bump_weight_epoch()                         # what optimizer.step() does
    ops.repack_all()
    ops.join_side_streams()                         # (the data-gradient copies are re-packed on the weight-gradient stream)
    L = ops.lib()
    for (m, shp), x in zip(layers, xs):
        pk = m._packed
        assert pk.fwd_key == ops._wkey(m.weight) and pk.dgr_key == ops._wkey(m.weight)
        d = m.geom.plan(shp[0], (1,) + tuple(shp[2:]))[0]
        ref_f, ref_d = torch.full_like(pk.fwd, 7.0), torch.full_like(pk.dgr, 7.0)
        got_f, got_d = pk.fwd.clone(), pk.dgr.clone()
        ops._ck(L.muvo_conv_pack_weights(C.byref(d), ops._f(m.weight), ops._f(ref_f), ops._f(ref_d), ops._st()))
        # padding the per-layer kernels never write keeps the 7.0 marker: compare where they wrote
        wf, wd = ref_f.view(torch.int32) != torch.tensor(7.0).view(torch.int32).item(), ref_d.view(torch.int32) != torch.tensor(7.0).view(torch.int32).item()
        assert torch.equal(got_f.view(torch.int32)[wf], ref_f.view(torch.int32)[wf]), type(m).__name__
        assert torch.equal(got_d.view(torch.int32)[wd], ref_d.view(torch.int32)[wd]), type(m).__name__
    pk = lin.weight._bf3_packed
    ref_f, ref_d = torch.empty_like(pk.fwd), torch.empty_like(pk.dgr)
    ops._ck(L.muvo_linear_bf16x3_pack(384, 1152, ops._f(lin.weight), ops._f(ref_f), ops._f(ref_d), ops._st()))
    assert torch.equal(pk.fwd.view(torch.int32), ref_f.view(torch.int32)) and torch.equal(pk.dgr.view(torch.int32), ref_d.view(torch.int32))


def test_seed_convt_as_gemm(dev):
    from muvo_amd.models.common import _Seed1x1ConvTFn
    from muvo_amd import ops
    torch.manual_seed(2)
    n, ci, co, kh, kw = 4, 48, 20, 5, 13
    w = (torch.randn(ci, co, kh, kw) * 0.1)
    b = torch.randn(co) * 0.1
    x = torch.randn(n, ci, 1, 1)
    wg = torch.nn.Parameter(w.to(dev))
    bg = torch.nn.Parameter(b.to(dev))
    wg.grad, bg.grad = torch.zeros_like(wg), torch.zeros_like(bg)
    xg = x.to(dev).requires_grad_(True)
    y = _Seed1x1ConvTFn.apply(xg, wg, bg, ops.ACT_ELU)
    wc, bc, xc = w.clone().requires_grad_(True), b.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yr = F.elu(F.conv_transpose2d(xc, wc, bc))
    _close(y, yr, name='seed fwd')
    g = torch.randn_like(yr)
    yr.backward(g)
    y.backward(g.to(dev))
    _close(xg.grad, xc.grad, name='seed dx')
    _close(wg.grad, wc.grad, name='seed dW')
    _close(bg.grad, bc.grad, name='seed db')


@pytest.mark.parametrize('res_mode,relu', [(0, True), (0, False), (1, True), (2, True)])
def test_batchnorm(dev, res_mode, relu):
    _batchnorm_case(dev, res_mode, relu, (5, 24, 9, 14))        # 126 elements per row: scalar kernels
    _batchnorm_case(dev, res_mode, relu, (3, 10, 32, 40))       # 1280 per row, % 4 == 0: float4 kernels


def _batchnorm_case(dev, res_mode, relu, shape):
    from muvo_amd import nn as hnn
    torch.manual_seed(3)
    n, c, h, w = shape
    with torch.device(dev):
        bn = hnn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(n, c, h, w) * 2 + 0.5
    r = torch.randn(n, c, h, w)
    xg = x.to(dev).requires_grad_(True)
    rg = r.to(dev).requires_grad_(True)
    bn.weight.grad, bn.bias.grad = torch.zeros_like(bn.weight), torch.zeros_like(bn.bias)
    y = bn(xg, residual=rg if res_mode else None, res_mode=res_mode or 1, relu=relu)
    ref = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        ref.weight.copy_(bn.weight.cpu())
        ref.bias.copy_(bn.bias.cpu())
    ref.train()
    xc, rc = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    yr = ref(xc)
    if res_mode == 1:
        yr = yr + rc
    if relu:
        yr = F.relu(yr)
    if res_mode == 2:
        yr = yr + rc
    _close(y, yr, name='bn fwd')
    _close(bn.running_mean, ref.running_mean, name='running_mean')
    _close(bn.running_var, ref.running_var, name='running_var')
    g = torch.randn_like(yr)
    yr.backward(g)
    y.backward(g.to(dev))
    _close(xg.grad, xc.grad, rtol=5e-4, name='bn dx')
    _close(bn.weight.grad, ref.weight.grad, rtol=5e-4, name='bn dgamma')
    _close(bn.bias.grad, ref.bias.grad, rtol=5e-4, name='bn dbeta')
    if res_mode:
        _close(rg.grad, rc.grad, name='bn dres')
    bn(x.to(dev))
    assert int(bn.state_dict()['num_batches_tracked']) == 2      # increments are applied lazily, before the buffer is read


def test_adain(dev):
    from muvo_amd import ops
    torch.manual_seed(4)
    n, c = 3, 10
    for shape, bcast in (((n, c, 6, 6, 4), False), ((c, 3, 3, 1), True), ((n, c, 16, 16, 8), False)):   # last: float4 kernels
        x = torch.randn(*shape) + 0.3
        style = torch.randn(n, 2 * c)
        xg, sg = x.to(dev).requires_grad_(True), style.to(dev).requires_grad_(True)
        y = ops.adain(xg, sg, 1e-8, n)
        xc, sc = x.clone().requires_grad_(True), style.clone().requires_grad_(True)
        xx = xc.unsqueeze(0).repeat(n, 1, 1, 1, 1) if bcast else xc
        mean = xx.mean(dim=(-1, -2, -3), keepdim=True)
        xm = xx - mean
        std = torch.sqrt((xm ** 2).mean(dim=(-1, -2, -3), keepdim=True) + 1e-8)
        yr = sc[:, :c, None, None, None] * (xm / std) + sc[:, c:, None, None, None]
        _close(y, yr, name='adain fwd')
        g = torch.randn_like(yr)
        yr.backward(g)
        y.backward(g.to(dev))
        _close(xg.grad, xc.grad, rtol=5e-4, name='adain dx')
        _close(sg.grad, sc.grad, rtol=5e-4, name='adain dstyle')


def test_conv_lrelu_adain_fused_backward(dev):
    """ConvInstanceNorm3d (common.py:190-202): conv + LeakyReLU(0.2) + AdaIN with the LeakyReLU derivative chained inside
    the AdaIN backward kernel (no separate activation-gradient pass) against the plain PyTorch composition."""
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    torch.manual_seed(11)
    n, cin, cout = 2, 16, 8
    with torch.device(dev):
        m = hnn.Conv3d(cin, cout, 3, 1, 1)
    x = torch.randn(n, cin, 4, 6, 32)
    style = torch.randn(n, 2 * cout)
    xg, sg = x.to(dev).requires_grad_(True), style.to(dev).requires_grad_(True)
    m.weight.grad, m.bias.grad = torch.zeros_like(m.weight), torch.zeros_like(m.bias)
    y = ops.adain(m(xg, act=ops.ACT_LEAKY, slope=0.2, act_bwd_fused=True), sg, 1e-8, n, ops.ACT_LEAKY, 0.2)
    w, b = m.weight.detach().cpu().requires_grad_(True), m.bias.detach().cpu().requires_grad_(True)
    xc, sc = x.clone().requires_grad_(True), style.clone().requires_grad_(True)
    h = F.leaky_relu(F.conv3d(xc, w, b, 1, 1), 0.2)
    hm = h - h.mean(dim=(-1, -2, -3), keepdim=True)
    yr = sc[:, :cout, None, None, None] * (hm / torch.sqrt((hm ** 2).mean(dim=(-1, -2, -3), keepdim=True) + 1e-8)) + \
        sc[:, cout:, None, None, None]
    _close(y, yr, name='fwd')
    g = torch.randn_like(yr)
    yr.backward(g)
    y.backward(g.to(dev))
    _close(xg.grad, xc.grad, rtol=5e-4, name='dx')
    _close(m.weight.grad, w.grad, rtol=5e-4, name='dW')
    _close(m.bias.grad, b.grad, rtol=5e-4, atol=1e-4, name='db')
    _close(sg.grad, sc.grad, rtol=5e-4, name='dstyle')


@pytest.mark.parametrize('pre', [False, True])
def test_adain_head_fused(dev, pre):
    """AdaIN + 1x1x1 class head in one pass (VoxelDecoder1's last stage, common.py:541-545 + :354-367): logits, the data /
    style gradients and the head's weight / bias gradients against the plain PyTorch composition; `pre`: the LeakyReLU
    derivative of the producing convolution chained inside the backward."""
    from muvo_amd import ops
    torch.manual_seed(21)
    n, c, co, shape = 3, 8, 2, (8, 12, 16)
    assert ops.lib().muvo_adain_head_supported(c, co, 8 * 12 * 16)
    z = torch.randn(n, c, *shape) + 0.3
    x = F.leaky_relu(z, 0.2) if pre else z
    style = torch.randn(n, 2 * c)
    hw, hb = torch.randn(co, c, 1, 1, 1) * 0.3, torch.randn(co)
    xg, sg = x.to(dev).requires_grad_(True), style.to(dev).requires_grad_(True)
    hwg, hbg = torch.nn.Parameter(hw.to(dev)), torch.nn.Parameter(hb.to(dev))
    hwg.grad, hbg.grad = torch.full_like(hwg, 0.5), torch.full_like(hbg, -0.25)       # the kernels ACCUMULATE into these
    xd = xg.detach().double()
    moments = torch.stack([xd.sum(dim=(2, 3, 4)), (xd * xd).sum(dim=(2, 3, 4))], dim=-1).contiguous()
    y = ops.adain_head(xg, sg, hwg, hbg, 1e-8, moments, ops.ACT_LEAKY if pre else ops.ACT_NONE, 0.2)
    assert float(moments.abs().max()) == 0.0                                           # handed back cleared
    zc, sc = z.clone().requires_grad_(True), style.clone().requires_grad_(True)
    hwc, hbc = hw.clone().requires_grad_(True), hb.clone().requires_grad_(True)
    h = F.leaky_relu(zc, 0.2) if pre else zc
    hm = h - h.mean(dim=(-1, -2, -3), keepdim=True)
    a = sc[:, :c, None, None, None] * (hm / torch.sqrt((hm ** 2).mean(dim=(-1, -2, -3), keepdim=True) + 1e-8)) + \
        sc[:, c:, None, None, None]
    yr = F.conv3d(a, hwc, hbc)
    _close(y, yr, name='logits')
    g = torch.randn_like(yr)
    yr.backward(g)
    y.backward(g.to(dev))
    _close(xg.grad, zc.grad, rtol=5e-4, name='dx')
    _close(sg.grad, sc.grad, rtol=5e-4, name='dstyle')
    _close(hwg.grad - 0.5, hwc.grad, rtol=5e-4, atol=1e-3, name='dhead_w')
    _close(hbg.grad + 0.25, hbc.grad, rtol=5e-4, atol=1e-3, name='dhead_b')


@pytest.mark.parametrize('cin,cout,shape', [(16, 8, (5, 13, 64)), (32, 16, (3, 9, 32)), (8, 8, (4, 7, 32))])
def test_decoder_block3d_lazy_adain(dev, cin, cout, shape):
    """DecoderBlock3d (common.py:161-202, 498-546): conv1 -> LeakyReLU -> AdaIN -> conv2 -> LeakyReLU -> AdaIN.  With the bf16x3
    voxel kernels conv2 applies the first AdaIN while staging its input (ops.adain_lazy, muvo_conv_forward_affine /
    muvo_conv_wgrad_affine): the normalised tensor is never written.  Output and every gradient against the plain PyTorch
    composition, and the fused path must really have been taken where it is supported."""
    from muvo_amd import ops
    from muvo_amd.models.common import DecoderBlock3d
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
    try:
        torch.manual_seed(9)
        lat = 24
        with torch.device(dev):
            blk = DecoderBlock3d(cin, cout, lat, upsample=False)
        n = 3
        x = torch.randn(n, cin, *shape)
        w = torch.randn(n, lat)
        xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        for p in blk.parameters():
            p.grad = torch.zeros_like(p)
        calls = []
        real = ops.adain_lazy
        ops.adain_lazy = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            y = blk(xg, wg)
        finally:
            ops.adain_lazy = real
        mid = torch.empty(n, cout, *shape, device=dev).requires_grad_(True)
        expect = ops.conv_affine_supported(mid, blk.conv2.conv_act[0].geom, ops.conv_moments_buffer(xg, blk.conv1.conv_act[0].geom))
        assert bool(calls) == bool(expect)
        if cin <= 16:
            assert expect, 'the 8 / 16-channel voxel layers must take the fused path'
        P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in blk.named_parameters()}
        xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)

        def cin3d(h, pre):
            h = F.leaky_relu(F.conv3d(h, P[pre + '.conv_act.0.weight'], P[pre + '.conv_act.0.bias'], padding=1), 0.2)
            st = F.linear(wc, P[pre + '.adaptive_norm.latent_affine.weight'], P[pre + '.adaptive_norm.latent_affine.bias'])
            c = h.shape[1]
            hm = h - h.mean(dim=(-1, -2, -3), keepdim=True)
            hn = hm / torch.sqrt((hm ** 2).mean(dim=(-1, -2, -3), keepdim=True) + 1e-8)
            return st[:, :c, None, None, None] * hn + st[:, c:, None, None, None]
        yr = cin3d(cin3d(xc, 'conv1'), 'conv2')
        _close(y, yr, rtol=5e-4, name='block output')
        g = torch.randn_like(yr)
        yr.backward(g)
        y.backward(g.to(dev))
        _close(xg.grad, xc.grad, rtol=1e-3, name='dx')
        _close(wg.grad, wc.grad, rtol=1e-3, atol=1e-3, name='dlatent')
        for k, v in blk.named_parameters():
            _close(v.grad, P[k].grad, rtol=1e-3, atol=2e-3, name='d' + k)
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)


@pytest.mark.parametrize('l,n,e', [(70, 3, 96), (324, 2, 384)])    # head dim 12: unfused attention; 48: the fused kernel
def test_transformer_layer(dev, l, n, e):
    from muvo_amd import nn as hnn
    torch.manual_seed(5)
    with torch.device(dev):
        layer = hnn.TransformerEncoderLayer(e, 8, dim_ff=160, dropout=0.0)
    ref = torch.nn.TransformerEncoderLayer(e, 8, dim_feedforward=160, dropout=0.0)
    ref.load_state_dict({k: v.cpu() for k, v in layer.state_dict().items()})
    ref.train()
    x = torch.randn(l, n, e)
    xg, xc = x.to(dev).requires_grad_(True), x.clone().requires_grad_(True)
    for p in layer.parameters():
        p.grad = torch.zeros_like(p)
    y = layer(xg, seed=1)
    yr = ref(xc)
    _close(y, yr, rtol=5e-4, name='transformer fwd')
    g = torch.randn_like(yr)
    yr.backward(g)
    y.backward(g.to(dev))
    _close(xg.grad, xc.grad, rtol=1e-3, name='transformer dx')
    refp = dict(ref.named_parameters())
    for k, p in layer.named_parameters():
        _close(p.grad, refp[k].grad, rtol=1e-3, atol=1e-4, name=f'transformer grad {k}')


def test_dropout_statistics(dev):
    from muvo_amd import ops
    x = torch.ones(1 << 20, device=dev, requires_grad=True)
    y = ops.dropout(x, 0.1, 1234)
    keep = (y > 0).float().mean().item()
    assert abs(keep - 0.9) < 5e-3
    assert abs(y.mean().item() - 1.0) < 1e-2
    y.sum().backward()
    assert torch.equal(x.grad, y.detach())  # same mask, same scale in backward
    y2 = ops.dropout(x, 0.1, 1235)
    assert not torch.equal(y2, y)


def test_tokens_roundtrip(dev):
    from muvo_amd import ops
    torch.manual_seed(6)
    n, c = 3, 40
    xi, xl = torch.randn(n, c, 5, 7), torch.randn(n, c, 2, 9)
    pi, pl = torch.randn(c, 35), torch.randn(c, 18)
    te = torch.randn(1, 1, c, 2)
    teg = torch.nn.Parameter(te.to(dev))
    teg.grad = torch.zeros_like(teg)
    xig, xlg = xi.to(dev).requires_grad_(True), xl.to(dev).requires_grad_(True)
    tok = ops.make_tokens(xig, xlg, pi.to(dev), pl.to(dev), teg)
    xic, xlc, tec = xi.clone().requires_grad_(True), xl.clone().requires_grad_(True), te.clone().requires_grad_(True)
    it = (xic + pi.view(1, c, 5, 7)).flatten(2).permute(2, 0, 1) + tec[:, :, :, 0]
    lt = (xlc + pl.view(1, c, 2, 9)).flatten(2).permute(2, 0, 1) + tec[:, :, :, 1]
    ref = torch.cat([it, lt], 0)
    _close(tok, ref, name='tokens')
    back_i = ops.untoken(tok, 0, 5, 7)
    back_l = ops.untoken(tok, 35, 2, 9)
    ri = ref[:35].permute(1, 2, 0).reshape(n, c, 5, 7)
    rl = ref[35:].permute(1, 2, 0).reshape(n, c, 2, 9)
    _close(back_i, ri, name='untoken img')
    _close(back_l, rl, name='untoken lidar')
    gi, gl = torch.randn_like(ri), torch.randn_like(rl)
    (ri * gi).sum().backward(retain_graph=True)
    (rl * gl).sum().backward()
    ((back_i * gi.to(dev)).sum() + (back_l * gl.to(dev)).sum()).backward()
    _close(xig.grad, xic.grad, name='tokens dxi')
    _close(xlg.grad, xlc.grad, name='tokens dxl')
    _close(teg.grad, tec.grad, rtol=5e-4, name='tokens dtype_emb')


def test_pooling_and_upsample(dev):
    from muvo_amd import ops
    torch.manual_seed(7)
    shapes = {'plain': torch.randn(2, 5, 13, 18), 'ties': (torch.randn(2, 3, 33, 70) * 2).round() / 2,   # many equal maxima
              'wide': torch.randn(1, 2, 64, 1024), 'tiny': torch.randn(3, 2, 2, 2),
              'ties8': (torch.randn(2, 3, 32, 72) * 2).round() / 2, 'stem': torch.randn(2, 4, 160, 416)}   # eight-column backward kernel
    for name, (k, s, p) in [(n_, c_) for n_ in shapes for c_ in ((3, 2, 1), (2, 2, 0))]:
        x = shapes[name]
        xg, xc = x.to(dev).requires_grad_(True), x.clone().requires_grad_(True)
        y, yr = ops.max_pool2d(xg, k, s, p), F.max_pool2d(xc, k, s, p)
        _close(y, yr, rtol=0, atol=0, name='maxpool fwd')
        g = torch.randn_like(yr)
        yr.backward(g)
        y.backward(g.to(dev))
        _close(xg.grad, xc.grad, rtol=0, atol=0, name=f'maxpool bwd {name} k{k}')
    x = shapes['plain']
    xg, xc = x.to(dev).requires_grad_(True), x.clone().requires_grad_(True)
    y, yr = ops.global_avg_pool(xg), xc.mean(dim=(-1, -2))
    _close(y, yr, name='avgpool')
    g = torch.randn_like(yr)
    yr.backward(g)
    y.backward(g.to(dev))
    _close(xg.grad, xc.grad, name='avgpool bwd')
    # vectorised (W % 4 == 0 / even) + scalar; W = 4, 8, 32: eight outputs per lane with neighbour columns from adjacent lanes
    for shape in ((2, 3, 3, 5, 2), (2, 3, 3, 5, 8), (1, 2, 4, 6, 4), (1, 1, 2, 3, 5), (1, 2, 3, 37, 32), (1, 1, 2, 3, 12), (1, 1, 6, 8, 12), (2, 1, 2, 2, 4),
                  (2, 3, 9, 32, 16), (1, 2, 17, 96, 32), (1, 1, 1, 48, 16)):      # the last three: the d-walking backward kernel
        v = torch.randn(*shape)
        vg, vc = v.to(dev).requires_grad_(True), v.clone().requires_grad_(True)
        y = ops.upsample3d_x2(vg)
        yr = F.interpolate(vc, scale_factor=2.0, mode='trilinear', align_corners=False)
        _close(y, yr, name=f'upsample3d fwd {shape}')
        g = torch.randn_like(yr)
        yr.backward(g)
        y.backward(g.to(dev))
        _close(vg.grad, vc.grad, name=f'upsample3d bwd {shape}')
    v1 = torch.randn(1, 2, 3, 3, 1)  # degenerate depth 1 (VoxelDecoder1 first levels)
    _close(ops.upsample3d_x2(v1.to(dev)), F.interpolate(v1, scale_factor=2.0, mode='trilinear', align_corners=False),
           name='upsample3d depth1')


def test_preprocess_kernels(dev):
    from muvo_amd import ops
    torch.manual_seed(8)
    img = torch.randint(0, 256, (1, 2, 3, 60, 96), dtype=torch.uint8)
    crop = (6, 14, 90, 46)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    label, norm = ops.preprocess_image(img.to(dev), crop, mean, std)
    ref = (img.float() / 255)[..., 14:46, 6:90]
    assert torch.equal(label.cpu(), ref)
    refn = (ref - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    _close(norm, refn, rtol=1e-6, atol=1e-6, name='normalise')
    l2 = ops.resize_bilinear(label, 16, 42)
    r2 = F.interpolate(ref.flatten(0, 1), size=(16, 42), mode='bilinear', align_corners=False).view(1, 2, 3, 16, 42)
    _close(l2, r2, rtol=1e-6, atol=1e-6, name='bilinear /2')
    route = torch.randint(0, 256, (1, 2, 3, 80, 80), dtype=torch.uint8)
    rn = ops.preprocess_route(route.to(dev), 64, mean, std)
    rr = F.interpolate((route.float() / 255).flatten(0, 1), size=(64, 64), mode='nearest').view(1, 2, 3, 64, 64)
    rr = (rr - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    _close(rn, rr, rtol=1e-6, atol=1e-6, name='route')
    rv = torch.randn(1, 2, 4, 8, 32)
    d = ops.divide_scalar(rv.to(dev), 50.0)
    assert torch.equal(d.cpu(), rv / 50.0)
    n2 = ops.resize_nearest(d, (4, 16))
    assert torch.equal(n2.cpu(), F.interpolate((rv / 50.0).flatten(0, 1), size=(4, 16), mode='nearest').view(1, 2, 4, 4, 16))
    vox = (torch.rand(1, 2, 1, 12, 12, 8) < 0.2).to(torch.uint8)
    v2 = ops.resize_nearest(vox.to(dev), (6, 6, 4))
    assert torch.equal(v2.cpu(), F.interpolate(vox.flatten(0, 1), size=(6, 6, 4), mode='nearest').view(1, 2, 1, 6, 6, 4))


def test_gru_and_sample(dev):
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    torch.manual_seed(9)
    b, h = 2, 96
    with torch.device(dev):
        cell = hnn.GRUCell(h, h)
    ref = torch.nn.GRUCell(h, h)
    ref.load_state_dict({k: v.cpu() for k, v in cell.state_dict().items()})
    x, h0 = torch.randn(b, h), torch.randn(b, h)
    xg, hg = x.to(dev).requires_grad_(True), h0.to(dev).requires_grad_(True)
    xc, hc = x.clone().requires_grad_(True), h0.clone().requires_grad_(True)
    for p in cell.parameters():
        p.grad = torch.zeros_like(p)
    y, yr = cell(xg, hg), ref(xc, hc)
    _close(y, yr, name='gru fwd')
    g = torch.randn_like(yr)
    yr.backward(g)
    y.backward(g.to(dev))
    _close(xg.grad, xc.grad, name='gru dx')
    _close(hg.grad, hc.grad, name='gru dh')
    for k, p in cell.named_parameters():
        _close(p.grad, dict(ref.named_parameters())[k].grad, rtol=5e-4, name=f'gru {k}')
    s = 40
    mls, eps = torch.randn(b, 2 * s), torch.randn(b, 3, 2, s)
    mg, mc = mls.to(dev).requires_grad_(True), mls.clone().requires_grad_(True)
    mu, sigma, sample = ops.rssm_sample(mg, eps.to(dev)[:, 1, 0], 0.1)
    rmu, rls = torch.split(mc, s, dim=-1)
    rsig = 2 * torch.sigmoid(rls / 2) + 0.1
    rsmp = rmu + rsig * eps[:, 1, 0]
    _close(mu, rmu, name='mu'); _close(sigma, rsig, name='sigma'); _close(sample, rsmp, name='sample')
    g1, g2, g3 = torch.randn(b, s), torch.randn(b, s), torch.randn(b, s)
    (rmu * g1 + rsig * g2 + rsmp * g3).sum().backward()
    (mu * g1.to(dev) + sigma * g2.to(dev) + sample * g3.to(dev)).sum().backward()
    _close(mg.grad, mc.grad, name='sample bwd')


def test_losses(dev):
    import sys
    from muvo_amd import ops
    from oracle import muvo_ref as R
    torch.manual_seed(10)
    b, s = 2, 3
    # spatial losses incl. ignore value and the partial-channel lidar form
    pred, tgt = torch.randn(b, s, 4, 6, 10), torch.randn(b, s, 4, 6, 10)
    tgt[0, 1, 0, 2, 3] = 255.0
    tgt[1, 0, 3, 1, 1] = 255.0
    pg, pc = pred.to(dev).requires_grad_(True), pred.clone().requires_grad_(True)
    out = ops.spatial_losses(pg, tgt.to(dev), [(0, 3, 2, 0.05), (3, 4, 1, 0.05)])
    r0 = 0.05 * R._spatial_regression(pc[:, :, :3], tgt[:, :, :3], 2)
    r1 = 0.05 * R._spatial_regression(pc[:, :, -1:], tgt[:, :, -1:], 1)
    _close(out[0], r0, name='lidar_re'); _close(out[1], r1, name='lidar_depth')
    (r0 * 1.5 + r1 * 0.5).backward()
    (out[0] * 1.5 + out[1] * 0.5).backward()
    _close(pg.grad, pc.grad, name='spatial bwd')
    # empty mask -> 0 (losses.py:91-92)
    t255 = torch.full_like(tgt, 255.0)
    z = ops.spatial_losses(pred.to(dev), t255.to(dev), [(0, 4, 1, 1.0)])
    assert z[0].item() == 0.0
    # voxel losses: 3 classes incl. an absent class and an ignore voxel
    logits = torch.randn(b, s, 3, 6, 5, 4)
    lab = torch.randint(0, 2, (b, s, 1, 6, 5, 4), dtype=torch.uint8)  # class 2 absent
    lg, lc = logits.to(dev).requires_grad_(True), logits.clone().requires_grad_(True)
    three = ops.voxel_losses(lg, lab.to(dev), 0.1)
    fl, ft = lc.flatten(0, 1), lab.flatten(0, 1)[:, 0]
    ce = 0.1 * F.cross_entropy(fl, ft.long(), reduction='none').mean()
    sem = 0.1 * R._sem_scal(fl, ft)
    geo = 0.1 * R._geo_scal(fl, ft)
    _close(three[0], ce, name='voxel ce'); _close(three[1], sem, name='sem_scal'); _close(three[2], geo, name='geo_scal')
    (ce * 1.0 + sem * 2.0 + geo * 3.0).backward()
    (three[0] * 1.0 + three[1] * 2.0 + three[2] * 3.0).backward()
    _close(lg.grad, lc.grad, rtol=5e-4, name='voxel bwd')
    lab255 = lab.clone(); lab255[0, 0, 0, 0, 0, 0] = 255
    t2 = ops.voxel_losses(logits.to(dev), lab255.to(dev), 1.0)
    _close(t2[1], R._sem_scal(logits.flatten(0, 1), lab255.flatten(0, 1)[:, 0]), name='sem_scal ignore')
    _close(t2[2], R._geo_scal(logits.flatten(0, 1), lab255.flatten(0, 1)[:, 0]), name='geo_scal ignore')
    # KL with the reference's first-step quirk, action L1
    pm, ps, qm, qs = (torch.randn(b, 4, 16) for _ in range(4))
    ps, qs = ps.abs() + 0.2, qs.abs() + 0.2
    gp = [t.to(dev).requires_grad_(True) for t in (pm, ps, qm, qs)]
    cp = [t.clone().requires_grad_(True) for t in (pm, ps, qm, qs)]
    kl = ops.kl_loss(*gp, 1e-3, 0.75)
    a = 0.75
    klr = 1e-3 * (a * R._kl(cp[0], cp[1], cp[2].detach(), cp[3].detach()) + (1 - a) * R._kl(cp[0].detach(), cp[1].detach(), cp[2], cp[3]))
    _close(kl[0], klr, name='kl')
    klr.backward(); kl[0].backward()
    for i, nme in enumerate(('pm', 'ps', 'qm', 'qs')):
        _close(gp[i].grad, cp[i].grad, name=f'kl d{nme}')
    p1, t1 = torch.randn(b, s, 1), torch.randn(b, s, 1)
    p1g, p1c = p1.to(dev).requires_grad_(True), p1.clone().requires_grad_(True)
    l1 = ops.l1_rows_loss(p1g, t1.to(dev), 1.0)
    l1r = (p1c - t1).abs().sum(-1, keepdim=True).mean()
    _close(l1[0], l1r, name='l1'); l1r.backward(); l1[0].backward()
    _close(p1g.grad, p1c.grad, name='l1 bwd')


def test_adamw_matches_torch(dev):
    from muvo_amd import ops
    torch.manual_seed(11)
    n = 10007
    p0, g = torch.randn(n), torch.randn(n)
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=3e-4, betas=(0.95, 0.999), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step in range(1, 4):
        pr.grad = g * step
        opt.step()
        ops.adamw_step(p, (g * step).to(dev), m, v, 3e-4, 0.95, 0.999, 1e-8, 0.01, step)
    _close(p, pr.data, rtol=1e-6, atol=1e-7, name='adamw')


def test_conv_strided_dgrad_uses_one_arithmetic_for_all_phases(dev):
    """Regression: the four sub-pixel phases of a 5x5 stride-2 data gradient have 9/6/6/4 taps; with a work threshold between
    their sizes only some qualified for bf16x3, and the fp32 phases then read dy without the ReLU derivative (the fused
    dy * act'(y) pass feeds only the bf16x3 kernels).  The plan now promotes all phases together."""
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.008)      # per-item phase work: 0.0106 (9 taps) ... 0.0047 (4 taps) GFLOP
    try:
        torch.manual_seed(0)
        with torch.device(dev):
            m = hnn.Conv2d(64, 64, 5, 2, 2)
        x = torch.randn(3, 64, 24, 24)
        xg = x.to(dev).requires_grad_(True)
        y = m(xg, act=ops.ACT_RELU)
        w, b = m.weight.detach().cpu().requires_grad_(True), m.bias.detach().cpu().requires_grad_(True)
        xc = x.clone().requires_grad_(True)
        yr = F.relu(F.conv2d(xc, w, b, 2, 2))
        g = torch.randn_like(yr)
        yr.backward(g)
        m.weight.grad, m.bias.grad = torch.zeros_like(m.weight), torch.zeros_like(m.bias)
        y.backward(g.to(dev))
        _close(y, yr, name='fwd')
        _close(xg.grad, xc.grad, name='dgrad')
        _close(m.weight.grad, w.grad, rtol=5e-4, name='wgrad')
        _close(m.bias.grad, b.grad, rtol=5e-4, name='dbias')
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)


@pytest.mark.parametrize('n,c,s', [(2, 70, 520), (2, 70, 521), (1, 8, 256), (3, 128, 1028), (1, 3, 300)])
def test_split_planes(dev, n, c, s):
    """The bf16x3 operand layout (muvo_split_planes): hi = bf16(x) (round to nearest even), lo = bf16(x - hi), channels-last with
    the channel count padded to 8 (zeros) — bit-exact; both kernel forms (S % 4 == 0 takes the 16-byte-load form); the
    backward-preamble variant multiplies by act'(y) first and adds per-channel sums to dbias."""
    import ctypes as C
    from muvo_amd import ops
    L = ops.lib()
    torch.manual_seed(n * 1000 + c + s)
    x = torch.randn(n, c, s, device=dev) * 3
    cp = (c + 7) // 8 * 8

    def run(xin, y=None, act=ops.ACT_NONE, slope=0.0, dbias=None):
        ws = torch.zeros((L.muvo_split_planes_bytes(n, c, C.c_int64(s)) + 3) // 4, device=dev, dtype=torch.float32)
        ops._ck(L.muvo_split_planes(ops._f(xin), ops._p(ws), n, c, C.c_int64(s), ops._f(y), act, C.c_float(slope), ops._f(dbias), ops._st()))
        planes = ws.view(torch.int16)[:2 * n * s * cp].view(2, n, s, cp)
        return planes[0].view(torch.bfloat16).float(), planes[1].view(torch.bfloat16).float()

    def ref(v):
        hi = v.to(torch.bfloat16).float()
        lo = (v - hi).to(torch.bfloat16).float()
        pad = lambda t: torch.nn.functional.pad(t.permute(0, 2, 1), (0, cp - c))
        return pad(hi), pad(lo)

    hi, lo = run(x)
    rhi, rlo = ref(x)
    assert torch.equal(hi, rhi) and torch.equal(lo, rlo)
    y = torch.randn(n, c, s, device=dev)
    db = torch.ones(c, device=dev)
    hi, lo = run(x, torch.nn.functional.leaky_relu(y, 0.2), ops.ACT_LEAKY, 0.2, db)
    z = x * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
    rhi, rlo = ref(z)
    assert torch.equal(hi, rhi) and torch.equal(lo, rlo)
    assert torch.allclose(db, 1 + z.sum((0, 2)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('l,n,heads,dh', [(324, 3, 8, 48), (70, 2, 4, 16), (33, 1, 2, 64), (16, 2, 1, 32), (384, 1, 2, 48), (1, 1, 1, 16)])
def test_flash_attention(dev, l, n, heads, dh):
    """csrc/attention.hip against (i) a plain PyTorch fp32 attention (no dropout) and (ii) the unfused GEMM + softmax_dropout +
    GEMM path with the same dropout seed (same mask index convention): output and all three input gradients."""
    from muvo_amd import ops
    assert ops.lib().muvo_attention_supported(l, dh) == 1
    torch.manual_seed(l * 7 + dh)
    e = heads * dh
    qkv = torch.randn(l, n, 3 * e, device=dev)
    g = torch.randn(l, n, e, device=dev)

    def run(fn, p, seed):
        x = qkv.clone().requires_grad_(True)
        o = fn(x, heads, p, seed)
        o.backward(g)
        return o.detach(), x.grad

    o, dq = run(ops.FlashAttentionFn.apply, 0.0, 0)
    x = qkv.clone().requires_grad_(True)
    q, k, v = (t.reshape(l, n, heads, dh).permute(1, 2, 0, 3) for t in x.split(e, dim=-1))
    att = torch.softmax(q @ k.transpose(-1, -2) / dh ** 0.5, dim=-1) @ v          # (n, heads, l, dh)
    ref = att.permute(2, 0, 1, 3).reshape(l, n, e)
    ref.backward(g)
    _close(o, ref, rtol=2e-5, name='flash attention fwd')
    _close(dq, x.grad, rtol=1e-4, name='flash attention dqkv')
    o2, dq2 = run(ops.FlashAttentionFn.apply, 0.0, 0)
    assert torch.equal(o, o2) and torch.equal(dq, dq2)                            # no atomics: bit-reproducible
    for p, seed in ((0.1, 77), (0.5, 123456789)):
        of, df = run(ops.FlashAttentionFn.apply, p, seed)
        ou, du = run(ops.AttentionFn.apply, p, seed)
        _close(of, ou, rtol=2e-5, name=f'flash vs unfused fwd p={p}')
        _close(df, du, rtol=1e-4, name=f'flash vs unfused dqkv p={p}')
        assert not torch.allclose(of, o)                                          # the mask does something


@pytest.mark.parametrize('b,t,dims', [(2, 5, (64, 32, 32, 8)), (1, 3, (64, 32, 32, 8)), (4, 4, (128, 64, 32, 16)), (2, 10, (1024, 512, 512, 64)),
                                      (6, 4, (64, 32, 32, 8)), (8, 12, (1024, 512, 512, 64))])   # > 4 sequences: slabs of 4 (BASELINE cfg 5: batch 8 x seq 12)
def test_fused_rssm(dev, b, t, dims):
    """The persistent RSSM kernels (csrc/rssm.hip) against the oracle's RSSM (plain torch on CPU; transition.py:76-173) and
    against the unfused per-op path of the same module: all eight outputs, the gradient w.r.t. the embedding and all 18
    parameter gradients, with the prior-sample branch taken mid-sequence and at the end, upstream gradients on every output."""
    from muvo_amd import ops
    from muvo_amd.models.transition import RSSM
    from oracle import muvo_ref as R
    H, S, E, A = dims
    torch.manual_seed(b * 100 + t + H)
    with torch.device(dev):
        m = RSSM(embedding_dim=E, action_dim=2, hidden_state_dim=H, state_dim=S, action_latent_dim=A, receptive_field=t,
                 use_dropout=True, dropout_probability=0.15)
    m.train()
    ref = R.RSSM(E, 2, H, S, A)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    emb, act = torch.randn(b, t, E), torch.rand(b, t, 2) * 2 - 1
    noise = torch.randn(b, t, 2, S)
    use_prior = [bool(i in (1, t - 1) and i > 0) for i in range(t)]
    keys = [(g, k) for g in ('prior', 'posterior') for k in ('hidden_state', 'sample', 'mu', 'sigma')]
    ups = {gk: torch.randn(b, t, H if gk[1] == 'hidden_state' else S) for gk in keys}

    def run(fused):
        ops.FUSED_RSSM = fused
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
        e = emb.to(dev).requires_grad_(True)
        out = m(e, act.to(dev), noise=noise.to(dev), use_prior=use_prior)
        sum((out[g][k] * ups[(g, k)].to(dev)).sum() for g, k in keys).backward()
        return out, e.grad, {n: p.grad.clone() for n, p in m.named_parameters()}

    assert ops.rssm_fused_supported(b, t, H, S, E, A, 2)
    try:
        out_f, de_f, gw_f = run(True)
        out_u, de_u, gw_u = run(False)
    finally:
        ops.FUSED_RSSM = True
    ec = emb.clone().requires_grad_(True)
    pri, pos = ref(ec, act, noise, use_prior)
    out_r = {'prior': pri, 'posterior': pos}
    sum((out_r[g][k] * ups[(g, k)]).sum() for g, k in keys).backward()
    tol = dict(rtol=2e-4, atol=2e-5) if H < 512 else dict(rtol=1e-3, atol=1e-4)
    for g, k in keys:
        _close(out_f[g][k], out_r[g][k], name=f'fused rssm {g}.{k}', **tol)
        _close(out_f[g][k], out_u[g][k], name=f'fused vs unfused {g}.{k}', **tol)
    _close(de_f, ec.grad, name='fused rssm d_embedding', **tol)
    _close(de_f, de_u, name='fused vs unfused d_embedding', **tol)
    refp = dict(ref.named_parameters())
    for n in gw_f:
        _close(gw_f[n], refp[n].grad, name=f'fused rssm grad {n}', **tol)
        _close(gw_f[n], gw_u[n], name=f'fused vs unfused grad {n}', **tol)


@pytest.mark.parametrize('shape', [(3, 64, 32, 40), (2, 70, 13, 20), (2, 12, 5, 13), (1, 128, 16, 256)])
@pytest.mark.parametrize('res_mode,relu', [(0, True), (0, False), (1, True), (2, True)])
def test_batchnorm_writes_split_planes(dev, shape, res_mode, relu):
    """muvo_bn_train_fwd_planes / muvo_bn_train_bwd_planes (round 4: the BatchNorm apply writes the consumer's bf16x3 operand
    format) against the unfused pair of passes - BatchNorm kernel, then muvo_split_planes over its result: the fp32 tensors, the
    hi / lo planes, the saved statistics, the running statistics and the parameter gradients must be BIT-identical (same
    arithmetic per element); both kernel forms (S % 4 == 0 and >= 1024: 16-byte loads; otherwise one pixel per lane); channel
    counts that are not multiples of 8 / 64; with y = NULL / dx = NULL (planes only) the planes are the same."""
    import ctypes as C
    from muvo_amd import ops
    L = ops.lib()
    n, c, h, w = shape
    s = h * w
    cp = (c + 7) // 8 * 8
    torch.manual_seed(c * 7 + s)
    x = (torch.randn(n, c, h, w, device=dev) * 2 + 0.5).contiguous()
    r = torch.randn(n, c, h, w, device=dev) if res_mode else None
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) - 0.5
    nws = (L.muvo_split_planes_bytes(n, c, C.c_int64(s)) + 3) // 4
    nplane = 2 * n * s * cp          # int16 elements of the two planes

    def split(t):
        ws = torch.zeros(nws, device=dev, dtype=torch.float32)
        ops._ck(L.muvo_split_planes(ops._f(t), ops._p(ws), n, c, C.c_int64(s), None, ops.ACT_NONE, C.c_float(0.0), None, ops._st()))
        return ws.view(torch.int16)[:nplane + 8].clone()        # + the zero page behind the planes

    def fwd(fused, want_y=True):
        y = torch.empty_like(x)
        mean, rstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        if fused:
            ws = torch.full((nws,), 7.0, device=dev)
            ops._ck(L.muvo_bn_train_fwd_planes(ops._f(x), ops._f(gamma), ops._f(beta), ops._f(r), ops._f(y) if want_y else None, ops._f(mean),
                                               ops._f(rstd), ops._f(rm), ops._f(rv), n, c, C.c_int64(s), C.c_float(1e-5), C.c_float(0.1),
                                               res_mode, int(relu), ops._p(ws), ops._st()))
            return y, ws.view(torch.int16)[:nplane + 8].clone(), mean, rstd, rm, rv
        st = torch.empty(2 * c, device=dev, dtype=torch.float64)
        ops._ck(L.muvo_bn_train_fwd(ops._f(x), ops._f(gamma), ops._f(beta), ops._f(r), ops._f(y), ops._f(mean), ops._f(rstd), ops._f(rm),
                                    ops._f(rv), ops._p(st), n, c, C.c_int64(s), C.c_float(1e-5), C.c_float(0.1), res_mode, int(relu), ops._st()))
        return y, split(y), mean, rstd, rm, rv
    was = ops.get_deterministic()
    ops.set_deterministic(True)          # one statistics workgroup per channel: the sums (and so every bit downstream) are reproducible
    try:
        ref = fwd(False)
        got = fwd(True)
        for a, b_, name in zip(got, ref, ('y', 'planes', 'mean', 'rstd', 'running_mean', 'running_var')):
            assert torch.equal(a, b_), f'forward {name}'
        assert torch.equal(fwd(True, want_y=False)[1], ref[1])
        y, mean, rstd = ref[0], ref[2], ref[3]
        dy = torch.randn(n, c, h, w, device=dev)
        mask_mode = 0 if not relu else (1 if res_mode == 1 else 2)

        def bwd(fused, want_dx=True):
            dx = torch.empty_like(x)
            dres = torch.empty_like(x) if res_mode == 1 else None
            dg, db = torch.ones(c, device=dev), torch.ones(c, device=dev)
            if fused:
                ws = torch.full((nws,), 7.0, device=dev)
                ops._ck(L.muvo_bn_train_bwd_planes(ops._f(x), ops._f(y), ops._f(dy), ops._f(gamma), ops._f(beta), ops._f(mean), ops._f(rstd),
                                                   ops._f(dx) if want_dx else None, ops._f(dres), ops._f(dg), ops._f(db), n, c, C.c_int64(s),
                                                   mask_mode, ops._p(ws), ops._st()))
                return dx, ws.view(torch.int16)[:nplane + 8].clone(), dres, dg, db
            st = torch.empty(2 * c, device=dev, dtype=torch.float64)
            ops._ck(L.muvo_bn_train_bwd(ops._f(x), ops._f(y), ops._f(dy), ops._f(gamma), ops._f(beta), ops._f(mean), ops._f(rstd), ops._f(dx),
                                        ops._f(dres), ops._f(dg), ops._f(db), ops._p(st), n, c, C.c_int64(s), mask_mode, ops._st()))
            return dx, split(dx), dres, dg, db
        ref = bwd(False)
        got = bwd(True)
        for a, b_, name in zip(got, ref, ('dx', 'planes', 'dres', 'dgamma', 'dbeta')):
            assert (a is None and b_ is None) or torch.equal(a, b_), f'backward {name}'
        assert torch.equal(bwd(True, want_dx=False)[1], ref[1])
    finally:
        ops.set_deterministic(was)


@pytest.mark.parametrize('shape,stride', [((4, 64, 40, 52), 1), ((3, 64, 32, 64), 2), ((2, 128, 10, 26), 1)])
def test_basic_block_with_planes_equals_without(dev, shape, stride):
    """BasicBlock -> BasicBlock (layers.py:9-66) with the BatchNorms writing split planes for the convolutions behind them and
    handing dx to the convolutions in front of them as planes (ops.BN_PLANES) against the same blocks with every tensor in fp32
    and separate split passes: outputs, input gradient and all parameter gradients bit-identical (deterministic mode), and the
    planes path is really taken (the inner activation exists as planes only)."""
    from muvo_amd import ops
    from muvo_amd.layers.layers import BasicBlock
    n, c, h, w = shape
    old_mode, was_det = ops.get_conv_mode(), ops.get_deterministic()
    ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=0.0)
    ops.set_deterministic(True)
    try:
        torch.manual_seed(5)
        with torch.device(dev):
            b1 = BasicBlock(c, c * stride, stride=stride, downsample=True if stride != 1 else None)
            b2 = BasicBlock(c * stride, c * stride)
        x0 = torch.randn(n, c, h, w, device=dev)
        g = None
        runs = []
        for planes in (False, True):
            ops.BN_PLANES = planes
            seen = []
            orig = ops.BNActFn.forward

            def spy(ctx, *a, _o=orig, _s=seen):
                _s.append((bool(a[6]), bool(a[7]), bool(a[8])))          # (planes, keep_f32, dx_planes)
                return _o(ctx, *a)
            ops.BNActFn.forward = staticmethod(spy)
            try:
                for p in list(b1.parameters()) + list(b2.parameters()):
                    p.grad = torch.zeros_like(p)
                x = x0.clone().requires_grad_(True)
                y = b2(b1(x, next_convs=(b2.conv1,)))
                if g is None:
                    g = torch.randn_like(y)
                y.backward(g)
                torch.cuda.synchronize()
            finally:
                ops.BNActFn.forward = staticmethod(orig)
            runs.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in list(b1.parameters()) + list(b2.parameters())], seen))
        (y0, dx0, g0, s0), (y1, dx1, g1, s1) = runs
        assert not any(p or d for p, _, d in s0)
        assert any(p and not k for p, k, _ in s1), s1        # bn1: planes only
        assert any(d for _, _, d in s1), s1                  # dx handed over as planes
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
        for a, b_ in zip(g0, g1):
            assert torch.equal(a, b_)
        assert torch.isfinite(dx1).all() and all(torch.isfinite(t).all() for t in g1)
    finally:
        ops.BN_PLANES = os.environ.get('MUVO_BN_PLANES', '1') != '0'
        ops.set_deterministic(was_det)
        ops.set_conv_mode(old_mode, min_gflop=-1.0)


@pytest.mark.parametrize('cin,shape', [(3, (2, 64, 832)), (3, (3, 38, 104)), (4, (2, 64, 1024)), (4, (1, 16, 424))])
def test_stem_convolution_kernel(dev, cin, shape):
    """The dedicated ResNet-18 stem kernels (csrc/conv_stem.hip: Conv2d(3 | 4, 64, 7, stride 2, padding 3, bias=False) on bf16x3
    products from an LDS-resident input patch; weight gradient with persistent accumulators) against PyTorch fp32 on the CPU:
    forward 2e-5 of the output scale, weight gradient 1e-4 of its scale (1e-5-sized bf16x3 rounding over 10^5 ... 10^6 pixel sums);
    through the module in the bf16x3 mode (the path ConvFn takes when the input needs no gradient), several tiles per row, a
    ragged last tile, odd row counts."""
    import ctypes as C
    from muvo_amd import nn as hnn
    from muvo_amd import ops
    n, h, w = shape
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_BF16X3, min_gflop=-1.0)
    try:
        torch.manual_seed(cin * 100 + h)
        with torch.device(dev):
            m = hnn.Conv2d(cin, 64, 7, 2, 3, bias=False)
        x = torch.randn(n, cin, h, w)
        m.weight.grad = torch.zeros_like(m.weight)
        y = m(x.to(dev))
        d = m.geom.plan(n, (1, h, w))[0]
        assert ops.lib().muvo_stem_conv_supported(C.byref(d)) == 1
        assert m.geom.family[('stem', n, (1, h, w), ops._plan_epoch[0])] is True
        wc = m.weight.detach().cpu().clone().requires_grad_(True)
        yr = F.conv2d(x, wc, None, 2, 3)
        assert y.shape == yr.shape
        err = (y.cpu() - yr).abs().max().item()
        assert err < 2e-5 * yr.abs().max().item(), (err, yr.abs().max().item())
        g = torch.randn_like(yr)
        yr.backward(g)
        y.backward(g.to(dev))
        ops.join_side_streams()
        gerr = (m.weight.grad.cpu() - wc.grad).abs().max().item()
        assert gerr < 1e-4 * wc.grad.abs().max().item(), (gerr, wc.grad.abs().max().item())
        # accumulation: a second backward adds
        y2 = m(x.to(dev))
        y2.backward(g.to(dev))
        ops.join_side_streams()
        assert (m.weight.grad.cpu() - 2 * wc.grad).abs().max().item() < 2e-4 * wc.grad.abs().max().item()
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)
