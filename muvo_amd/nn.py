"""HIP-backed layer modules.  They are torch.nn.Module subclasses only to own parameters/buffers under the
reference's state_dict names; every forward goes through muvo_amd.ops (hand-written gfx950 kernels).
Initialisation mirrors PyTorch's defaults for the corresponding torch.nn layers (own restatement)."""
import math

import torch
import torch.nn as nn

from muvo_amd import ops


def _uniform_(t, bound):
    with torch.no_grad():
        t.uniform_(-bound, bound)


class _ConvNd(nn.Module):
    def __init__(self, nd, transposed, cin, cout, ksz, stride=1, pad=0, dil=1, out_pad=0, bias=True):
        super().__init__()
        self.geom = ops.ConvGeom(nd, transposed, cin, cout, ksz, stride, pad, dil, out_pad)
        k = self.geom.ksz[3 - nd:]
        shape = (cin, cout, *k) if transposed else (cout, cin, *k)
        self.weight = nn.Parameter(torch.empty(shape))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self._packed = ops._PackedWeights()
        fan_in = shape[1] * math.prod(k)
        _uniform_(self.weight, 1.0 / math.sqrt(fan_in))  # kaiming_uniform(a=sqrt(5)) bound
        if bias:
            _uniform_(self.bias, 1.0 / math.sqrt(fan_in))

    def forward(self, x, act=ops.ACT_NONE, slope=0.0, act_bwd_fused=False, moments=None, lazy=None):
        return ops.conv(x, self.weight, self.bias, self.geom, self._packed, act, slope, act_bwd_fused, moments, lazy)


class Conv2d(_ConvNd):
    def __init__(self, cin, cout, ksz, stride=1, pad=0, bias=True):
        super().__init__(2, False, cin, cout, ksz, stride, pad, 1, 0, bias)


class Conv3d(_ConvNd):
    def __init__(self, cin, cout, ksz, stride=1, pad=0, bias=True):
        super().__init__(3, False, cin, cout, ksz, stride, pad, 1, 0, bias)


class ConvTranspose2d(_ConvNd):
    def __init__(self, cin, cout, ksz, stride=1, pad=0, out_pad=0, bias=True):
        super().__init__(2, True, cin, cout, ksz, stride, pad, 1, out_pad, bias)


class Linear(nn.Module):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        _uniform_(self.weight, 1.0 / math.sqrt(cin))
        if bias:
            _uniform_(self.bias, 1.0 / math.sqrt(cin))

    def forward(self, x, act=ops.ACT_NONE, slope=0.0):
        return ops.linear(x, self.weight, self.bias, act, slope)


class BatchNorm2d(nn.Module):
    """Parameter/buffer holder; the compute is ops.bn_act (always batch statistics, like the reference which keeps
    train() mode for validation too)."""

    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, x, residual=None, res_mode=1, relu=False, consumers=(), sole_consumer=False, from_conv=False):
        """consumers / sole_consumer / from_conv: see ops.bn_act (the result written as the consumers' split planes, dx handed to
        the producing convolution as planes)"""
        return ops.bn_act(x, self, residual, res_mode, relu, consumers, sole_consumer, from_conv)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        ops.flush_bn_counters()          # num_batches_tracked increments are applied lazily, in one fused launch
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        ops.flush_bn_counters()          # pending increments belong to the values that are about to be overwritten
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class LayerNorm(nn.Module):
    def __init__(self, e, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(e))
        self.bias = nn.Parameter(torch.zeros(e))


class Placeholder(nn.Module):
    """Parameter-free slot (activation positions inside reference nn.Sequential containers) so that child indices
    — and therefore state_dict names — match the reference."""

    def forward(self, x):
        return x


class _MHAParams(nn.Module):
    def __init__(self, e):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * e, e))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * e))
        self.out_proj = Linear(e, e)
        bound = math.sqrt(6.0 / (e + 3 * e))  # xavier_uniform
        _uniform_(self.in_proj_weight, bound)
        with torch.no_grad():
            self.out_proj.bias.zero_()


class TransformerEncoderLayer(nn.Module):
    """nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward=2048, dropout, relu, post-LN, seq-first)
    (mile.py:96-101).  Dropout sites: attention probabilities, after out_proj (dropout1), inside FFN (dropout),
    after linear2 (dropout2)."""

    def __init__(self, e, nhead, dim_ff=2048, dropout=0.1):
        super().__init__()
        self.nhead, self.p = nhead, dropout
        self.self_attn = _MHAParams(e)
        self.linear1 = Linear(e, dim_ff)
        self.linear2 = Linear(dim_ff, e)
        self.norm1 = LayerNorm(e)
        self.norm2 = LayerNorm(e)

    def forward(self, x, seed):
        p = self.p if self.training else 0.0
        # validation (trainer.py:404-409) switches only the nn.Dropout MODULES off; MHA's functional dropout stays on
        pm = 0.0 if getattr(self, 'module_dropout_off', False) else p
        sa = self.self_attn
        qkv = ops.linear(x, sa.in_proj_weight, sa.in_proj_bias)
        o = ops.attention(qkv, self.nhead, p, seed)
        a = sa.out_proj(o)
        x = ops.add_dropout_layernorm(x, a, self.norm1, pm, seed + 1)
        f = self.linear1(x, act=ops.ACT_RELU)
        f = ops.dropout(f, pm, seed + 2)
        f = self.linear2(f)
        return ops.add_dropout_layernorm(x, f, self.norm2, pm, seed + 3)


class TransformerEncoder(nn.Module):
    def __init__(self, e, nhead, num_layers, dropout=0.1):
        super().__init__()
        self.layers = nn.ModuleList(TransformerEncoderLayer(e, nhead, dropout=dropout) for _ in range(num_layers))

    def forward(self, x, seed):
        for i, layer in enumerate(self.layers):
            x = layer(x, seed + 16 * i)
        return x


class GRUCell(nn.Module):
    """nn.GRUCell parameters (weight_ih, weight_hh, bias_ih, bias_hh) + MFMA GEMMs + fused pointwise kernel."""

    def __init__(self, cin, hid):
        super().__init__()
        self.hid = hid
        b = 1.0 / math.sqrt(hid)
        self.weight_ih = nn.Parameter(torch.empty(3 * hid, cin))
        self.weight_hh = nn.Parameter(torch.empty(3 * hid, hid))
        self.bias_ih = nn.Parameter(torch.empty(3 * hid))
        self.bias_hh = nn.Parameter(torch.empty(3 * hid))
        for p in self.parameters():
            _uniform_(p, b)

    def forward(self, x, h):
        gi = ops.linear(x, self.weight_ih, self.bias_ih)
        gh = ops.linear(h, self.weight_hh, self.bias_hh)
        return ops.gru_pointwise(gi, gh, h)
