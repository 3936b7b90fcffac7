"""ctypes binding of libmuvo_hip.so (include/muvo_hip.h) + torch.autograd glue.

PyTorch is used here only as plumbing: device memory (torch.empty), the current HIP stream, and the
autograd tape.  Every arithmetic op of the training step is a call into the hand-written gfx950
kernels; there is NO eager/CPU fallback — if the shared library is missing or a call fails, a
RuntimeError is raised (the product path must fail loudly).

Parameter gradients are accumulated by the kernels straight into `param.grad` (pre-allocated, usually a
view into one flat gradient buffer, see muvo_amd/param_store.py); the autograd Functions therefore
return None for parameters and only propagate activation gradients.
"""
import contextlib
import ctypes as C
import os
import threading
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get('MUVO_HIP_LIB') or os.path.join(_HERE, 'libmuvo_hip.so')   # MUVO_HIP_LIB: A/B builds on the GPU box
_lib = None
_lock = threading.Lock()

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_ELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4, 5


class ConvDesc(C.Structure):
    _fields_ = [('nd', C.c_int32), ('transposed', C.c_int32), ('N', C.c_int32), ('Cin', C.c_int32), ('Cout', C.c_int32),
                ('in_sz', C.c_int32 * 3), ('out_sz', C.c_int32 * 3), ('ksz', C.c_int32 * 3), ('stride', C.c_int32 * 3),
                ('pad', C.c_int32 * 3), ('dil', C.c_int32 * 3)]


class GemmDesc(C.Structure):
    _fields_ = [('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('sam', C.c_int64), ('sak', C.c_int64), ('sbk', C.c_int64), ('sbn', C.c_int64), ('scm', C.c_int64),
                ('B1', C.c_int32), ('B2', C.c_int32),
                ('a_b1', C.c_int64), ('a_b2', C.c_int64), ('b_b1', C.c_int64), ('b_b2', C.c_int64),
                ('c_b1', C.c_int64), ('c_b2', C.c_int64),
                ('alpha', C.c_float), ('bias_div', C.c_int32), ('act', C.c_int32), ('slope', C.c_float),
                ('mode', C.c_int32)]


# every exported symbol of include/muvo_hip.h (tests/test_abi.py checks this list against the header)
EXPORTS = [
    'muvo_last_error', 'muvo_abi_version', 'muvo_selftest_mfma', 'muvo_set_deterministic', 'muvo_get_deterministic', 'muvo_reset_accumulators',
    'muvo_conv_set_products', 'muvo_conv_get_products',
    'muvo_conv_set_mode', 'muvo_conv_get_mode', 'muvo_conv_set_bf16x3_min_gflop', 'muvo_conv_pack_sizes', 'muvo_conv_workspace_bytes', 'muvo_conv_kernel_family', 'muvo_conv_kernel_variant', 'muvo_conv_pack_weights', 'muvo_conv_forward', 'muvo_conv_dgrad', 'muvo_conv_prepare_dy', 'muvo_conv_wgrad', 'muvo_bias_grad_nchw',
    'muvo_gemm',
    'muvo_pack_table_item_bytes', 'muvo_conv_pack_table_add', 'muvo_linear_bf16x3_pack_table_add', 'muvo_pack_table_run',
    'muvo_linear_bf16x3_pack_floats', 'muvo_linear_bf16x3_pack', 'muvo_linear_bf16x3_workspace_bytes', 'muvo_linear_bf16x3_split',
    'muvo_linear_bf16x3_forward', 'muvo_linear_bf16x3_dgrad', 'muvo_linear_bf16x3_wgrad',
    'muvo_bn_train_fwd', 'muvo_bn_train_bwd', 'muvo_adain_fwd', 'muvo_adain_bwd',
    'muvo_add_dropout_layernorm_fwd', 'muvo_add_dropout_layernorm_bwd',
    'muvo_act_fwd', 'muvo_act_bwd', 'muvo_dropout', 'muvo_axpby', 'muvo_copy2d', 'muvo_colsum_acc', 'muvo_batchsum',
    'muvo_nchw_to_tokens', 'muvo_tokens_to_nchw', 'muvo_maxpool2d_fwd', 'muvo_maxpool2d_bwd', 'muvo_avgpool_fwd',
    'muvo_avgpool_bwd', 'muvo_upsample3d_x2_fwd', 'muvo_upsample3d_x2_bwd', 'muvo_preprocess_image',
    'muvo_preprocess_route', 'muvo_divide_scalar', 'muvo_resize_bilinear', 'muvo_resize_nearest_f32',
    'muvo_resize_nearest_u8', 'muvo_softmax_dropout_fwd', 'muvo_softmax_dropout_bwd', 'muvo_gru_fwd', 'muvo_gru_bwd',
    'muvo_rssm_sample_fwd', 'muvo_rssm_sample_bwd',
    'muvo_spatial_loss_fwd', 'muvo_spatial_loss_bwd', 'muvo_spatial_loss_masked_fwd', 'muvo_spatial_loss_masked_bwd', 'muvo_voxel_loss_stats_doubles', 'muvo_voxel_loss_coef_floats',
    'muvo_voxel_loss_fwd', 'muvo_voxel_loss_bwd', 'muvo_l1_rows_fwd', 'muvo_l1_rows_bwd', 'muvo_kl_loss_fwd',
    'muvo_kl_loss_bwd', 'muvo_adamw_step',
    'muvo_ssim_frames', 'muvo_sqdiff_frames', 'muvo_chamfer_sums', 'muvo_ssc_counts',
    'muvo_frustum_cells', 'muvo_frustum_pool_fwd', 'muvo_frustum_pool_bwd', 'muvo_depth_expectation',
    'muvo_resize_bilinear_bwd', 'muvo_softmax_channel_fwd', 'muvo_softmax_channel_bwd',
    'muvo_range_projection', 'muvo_voxel_grid', 'muvo_seg_ce_fwd', 'muvo_seg_ce_bwd', 'muvo_ssim_maps', 'muvo_ssim_bwd',
    'muvo_instance_labels', 'muvo_pixel_augment', 'muvo_preprocess_route_aug',
    'muvo_split_planes_bytes', 'muvo_split_planes', 'muvo_attention_supported', 'muvo_attention_fwd', 'muvo_attention_bwd',
    'muvo_conv_dgrad_accumulate', 'muvo_conv_forward_moments_supported', 'muvo_conv_forward_moments', 'muvo_adain_fwd_moments',
    'muvo_adain_head_supported', 'muvo_adain_head_fwd', 'muvo_adain_head_bwd', 'muvo_bf3_loop_clock',
    'muvo_grouped_linear_fwd', 'muvo_grouped_linear_bwd', 'muvo_conv_prepare_dy_head', 'muvo_conv_prepare_dy_head_supported',
    'muvo_conv_forward_head_supported', 'muvo_conv_forward_head',
    'muvo_adain_affine', 'muvo_conv_affine_supported', 'muvo_conv_forward_affine', 'muvo_conv_wgrad_affine',
    'muvo_fake_allreduce', 'muvo_resize_bilinear_aa', 'muvo_bn_train_fwd_planes', 'muvo_bn_train_bwd_planes', 'muvo_conv_forward_planes',
    'muvo_stem_conv_supported', 'muvo_stem_conv_forward', 'muvo_stem_conv_wgrad',
    'muvo_rssm_supported', 'muvo_rssm_transposed_floats', 'muvo_rssm_scratch_floats', 'muvo_rssm_forward', 'muvo_rssm_backward',
]


def lib():
    """Load the shared library (once). Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(_LIB_PATH):
                    raise RuntimeError(f'{_LIB_PATH} not found: build it with `python -m muvo_amd.build` '
                                       '(there is no fallback path)')
                L = C.CDLL(_LIB_PATH)
                L.muvo_last_error.restype = C.c_char_p
                L.muvo_conv_workspace_bytes.restype = C.c_int64
                L.muvo_linear_bf16x3_workspace_bytes.restype = C.c_int64
                L.muvo_pack_table_item_bytes.restype = C.c_int64
                L.muvo_split_planes_bytes.restype = C.c_int64
                L.muvo_rssm_transposed_floats.restype = C.c_int64
                L.muvo_rssm_scratch_floats.restype = C.c_int64
                for name in EXPORTS:
                    getattr(L, name)  # AttributeError if a declared symbol is missing
                _lib = L
    return _lib


def _ck(rc):
    if rc != 0:
        msg = lib().muvo_last_error().decode()
        reset_accumulators()
        raise RuntimeError(f'muvo_hip error {rc}: {msg}')


_MOMENT_BUFS = []      # weak references to the per-layer float64 moments buffers (conv_moments_buffer)


def reset_accumulators():
    """Error path: buffers that are all-zero between uses by construction - the library's per-stream statistics accumulators and
    the per-layer moments buffers a convolution epilogue fills for the AdaIN that follows - may hold partial sums when an
    exception interrupted a step between producer and consumer.  Called by _ck on every library error and by the trainer when
    a step raises; a no-op cost otherwise (never on the success path)."""
    try:
        if _lib is not None:
            _lib.muvo_reset_accumulators()
        live = []
        for r in _MOMENT_BUFS:
            t = r()
            if t is not None:
                t.zero_()
                live.append(r)
        _MOMENT_BUFS[:] = live
    except Exception:          # the original error is the one to report
        pass


def _st():
    # raw handle of the current PyTorch stream of the current device (fast path: one C call)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _p(t):
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous(), 'muvo_hip ops need contiguous device tensors'
    return C.c_void_p(t.data_ptr())


def _f(t):
    assert t is None or t.dtype == torch.float32
    return _p(t)


def _i64(v):
    return C.c_int64(int(v))


def _fl(v):
    return C.c_float(float(v))


_weight_epoch = [0]


def bump_weight_epoch():
    """Call after parameters were modified outside torch (the fused AdamW kernel) to invalidate packed weights."""
    _weight_epoch[0] += 1


CONV_F32, CONV_BF16X3, CONV_BF16 = 0, 1, 2     # CONV_BF16: the bf16x3 kernels with ONE product (extension, see include/muvo_hip.h)


def set_conv_mode(mode, min_gflop=None):
    """Select the matrix-pipe arithmetic of the large convolutions (CONV_F32 exact / CONV_BF16X3 split products;
    min_gflop = per-item work below which a phase stays on fp32 MFMA; negative = the library's built-in per-layer policy).  Packed weights and plans depend on it, so both
    caches are invalidated."""
    _ck(lib().muvo_conv_set_mode(CONV_BF16X3 if int(mode) == CONV_BF16 else int(mode)))
    _ck(lib().muvo_conv_set_products(1 if int(mode) == CONV_BF16 else 3))
    if min_gflop is not None:
        _ck(lib().muvo_conv_set_bf16x3_min_gflop(C.c_double(min_gflop)))
    bump_weight_epoch()
    _plan_epoch[0] += 1


def get_conv_mode():
    m = lib().muvo_conv_get_mode()
    return CONV_BF16 if (m == CONV_BF16X3 and lib().muvo_conv_get_products() == 1) else m


_plan_epoch = [0]


def set_deterministic(on=True):
    """Deterministic mode (include/muvo_hip.h: muvo_set_deterministic; MUVO_DETERMINISTIC=1 at start-up): every reduction whose
    result would depend on the arrival order of float / double atomics runs in a fixed order - two runs of the same step are
    bit-identical.  Slower (split-K and pixel-range splits off, ordered workgroup turns, the AdaIN style projections one Linear
    at a time); a debugging aid for parity work.  Plans and packed weights depend on it: both caches are invalidated."""
    global GROUPED_LINEAR
    _ck(lib().muvo_set_deterministic(1 if on else 0))
    GROUPED_LINEAR = (not on) and os.environ.get('MUVO_GROUPED_LINEAR', '1') != '0'   # its data gradient adds 14 layers atomically
    bump_weight_epoch()
    _plan_epoch[0] += 1


def get_deterministic():
    return bool(lib().muvo_get_deterministic())


_join_queued = [False]


def _queue_final_join():
    """Parameter gradients are written by kernels on whatever stream their backward node runs on (branch streams, the
    weight-gradient stream), not by AccumulateGrad nodes, so the autograd engine does not know about them: the first gradient
    write of a backward pass installs an engine callback that makes the stream `backward()` was called on wait for every
    side stream when the pass ends - `p.grad` is then safe to read right after `loss.backward()` like any other gradient."""
    def cb():
        _join_queued[0] = False
        join_side_streams()
    try:
        torch.autograd.Variable._execution_engine.queue_callback(cb)
        _join_queued[0] = True
    except RuntimeError:      # not inside a backward pass
        pass


def grad_of(p):
    """The gradient buffer the kernels accumulate into.  With a ParamStore that is the parameter's slot of the flat gradient
    buffer: if something set p.grad to None (nn.Module.zero_grad(), optimizer.zero_grad(set_to_none=True) of a foreign
    loop) the slot is zeroed and re-bound here, so gradients never land in a detached tensor.  Without a store the tensor is
    allocated on demand.  Frozen parameters (requires_grad=False, OPTIMIZER.FROZEN) get no buffer of their own: the kernels
    that always write a parameter gradient (BatchNorm / LayerNorm backward) add into one shared dump nobody reads."""
    if not p.requires_grad:
        return scratch('frozen_grad_dump', p.numel(), p.device).view(-1)[:p.numel()].view(p.shape)
    if _side_streams and not _join_queued[0]:
        _queue_final_join()
    if p.grad is None:
        view = getattr(p, '_muvo_flat_grad', None)
        if view is not None:
            view.zero_()
            p.grad = view
        else:
            p.grad = torch.zeros_like(p)
    return p.grad


_scratch = {}


def _stream_key(device):
    """Scratch buffers are per (device, HIP stream): independent branches of the model run on side streams (see `branch`) and
    must not share workspaces."""
    device = torch.device(device)
    if device.type != 'cuda':
        return 0
    return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())


def scratch_zeroed(name, nfloats, device):
    """Scratch that is all-zero at allocation; its users (weight-gradient kernels + unpack) leave it all-zero."""
    key = (name, device, torch.float32, _stream_key(device))
    t = _scratch.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.zeros(int(nfloats), device=device, dtype=torch.float32)
        _scratch[key] = t
    return t


def scratch(name, nfloats, device, dtype=torch.float32):
    key = (name, device, dtype, _stream_key(device))
    t = _scratch.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(int(nfloats), device=device, dtype=dtype)
        _scratch[key] = t
    return t


# ------------------------------------------------------------------------------------------------
# Independent sub-networks on side HIP streams.  The route-map encoder (a ResNet-18 on 64 x 64 pixels: ~300 launches of 5-50 us
# that keep a handful of compute units busy), the range-view encoder and two of the three decoders do not depend on what the
# main stream is doing at that time: issued on their own stream they fill the compute units the main stream's kernels leave
# idle.  autograd runs the backward of every node on the stream its forward ran on and synchronises across streams itself;
# parameter gradients are written by the kernels (not by AccumulateGrad nodes), so whoever reads them (optimizer, gradient
# exchange) first calls join_side_streams().  MUVO_STREAMS=0: everything on the current stream.
STREAMS = os.environ.get('MUVO_STREAMS', '1') != '0'
_side_streams = {}
_inputs_ready = {}


# Which logical branch runs on which HIP stream.  Measured on MI355X (profiles/r03i_stream_queues.txt): concurrency pays as long
# as the process keeps AT MOST FOUR streams busy (the current stream included) - a fifth concurrently active queue does not
# serialise gracefully, the step falls from 84 to 104-111 ms - and with the runtime's default of four hardware queues the
# streams of other libraries (RCCL) push ours onto shared queues in an order nobody controls.  So: muvo_amd/__init__.py asks for
# eight hardware queues (GPU_MAX_HW_QUEUES, before HIP initialises) so that every stream below owns one, and the logical
# branches are folded onto a BUDGET of side streams: three alone on the GPU ({encoders, then the range-view decoder} | voxel
# decoder | weight gradients), two when a gradient exchange is attached (its communication stream only carries waits, RCCL's
# own stream is the fourth busy one; the voxel decoder then follows the range-view decoder's stream).  Same-box A/Bs: range-view
# decoder on s0 instead of the main stream 83.3 -> 82.3 ms/step; with a one-rank RCCL group 84.7 -> 83.1.  MUVO_STREAM_MAP
# overrides single entries ("voxel_decoder=main,wgrad=s0").
# (History: the range-view decoder had to leave its side stream for a while - next to the RGB decoder single heads came out with
# a few 16-element groups off by ~1e-2 in about half of all processes.  That was the packed-fp32 instruction finding of
# muvo_amd/build.py; since the library is built without those instructions: 0 of 30 processes, profiles/r03j_lidar_decoder_stream.txt.)
# The ORDER in which the decoders are recorded in forward matters for backward, see models/mile.py.
_STREAM_PLANS = {
    3: {'lidar_encoder': 's0', 'route_encoder': 's0', 'lidar_decoder': 's0', 'voxel_decoder': 's1', 'wgrad': 's2'},
    2: {'lidar_encoder': 's0', 'route_encoder': 's0', 'lidar_decoder': 's0', 'voxel_decoder': 's0', 'wgrad': 's1'},
    1: {'lidar_encoder': 's0', 'route_encoder': 's0', 'lidar_decoder': 'main', 'voxel_decoder': 'main', 'wgrad': 's0'},
    0: {},
}
STREAM_BUDGET = [int(os.environ.get('MUVO_STREAM_BUDGET', '3'))]
STREAM_MAP = dict(kv.split('=') for kv in os.environ.get('MUVO_STREAM_MAP', '').split(',') if '=' in kv)


def set_stream_budget(n):
    """number of side streams the model may use (0..3); SegmentedGradReducer lowers it to 2 when a gradient exchange is attached
    (its communication stream only carries waits, RCCL's own stream is the fourth busy one)"""
    STREAM_BUDGET[0] = max(0, min(3, int(n)))


def side_stream(name, device):
    device = torch.device(device)
    plan = _STREAM_PLANS[max(0, min(3, STREAM_BUDGET[0]))]
    name = STREAM_MAP.get(name, plan.get(name, 'main' if name in _STREAM_PLANS[3] else name))
    if name == 'main':
        return torch.cuda.current_stream(device)
    key = (name, device.index)
    st = _side_streams.get(key)
    if st is None:
        # default: the stream that carries ONLY the weight gradients (leaves of the backward graph: nothing waits for them before
        # the optimizer) runs at low priority - the main stream's chain is the critical path of the step and gets the compute
        # units first (same-box A/Bs, profiles/r04a_priority.txt: 80.9 -> 79.9 and 80.6 -> 80.3 ms/step)
        dflt = 1 if (plan.get('wgrad') == name and list(plan.values()).count(name) == 1) else 0
        st = _side_streams[key] = _new_stream(device, SIDE_PRIORITY.get(name, SIDE_PRIORITY.get('*', dflt)))
    return st


def reserve_memory_pools(device, main_gb=None, side_gb=None):
    """Pre-size PyTorch's caching allocator for a training loop: one large block is allocated and released on the current stream
    and on every side stream that EXISTS (call it after a first step: creating the streams here, in another order, changes their
    hardware-queue assignment), so that each stream's pool serves the step's requests by splitting it.  Without this the pools grow
    while the loop runs - blocks are stream-specific and a block freed on one stream while a kernel of another still reads it
    (record_stream) comes back late, so the allocator keeps calling hipMalloc for dozens of steps.  MUVO_POOL_MAIN_GB (default 32) +
    MUVO_POOL_SIDE_GB (default 12) per side stream; returns the bytes reserved."""
    device = torch.device(device)
    if device.type != 'cuda':
        return 0
    main_gb = float(os.environ.get('MUVO_POOL_MAIN_GB', 32.0)) if main_gb is None else main_gb
    side_gb = float(os.environ.get('MUVO_POOL_SIDE_GB', 12.0)) if side_gb is None else side_gb
    cur = torch.cuda.current_stream(device)
    todo, seen = [(cur, main_gb)], {cur.cuda_stream}
    for (name, idx), st in sorted(_side_streams.items(), key=lambda kv: kv[0][0]):
        if idx == device.index and st.cuda_stream not in seen:
            seen.add(st.cuda_stream)
            todo.append((st, side_gb))
    free = torch.cuda.mem_get_info(device)[0]
    total = 0
    for st, gb in todo:
        nbytes = int(gb * 2 ** 30)
        if nbytes <= 0 or total + nbytes > 0.8 * free:
            continue
        with torch.cuda.stream(st):
            t = torch.empty(nbytes, dtype=torch.uint8, device=device)
            del t
        total += nbytes
    return total


# HIP stream priorities of the side streams (MUVO_SIDE_PRIORITY="*=normal" | "s2=low,s0=normal" ...; physical stream names s0..s2).
# The main stream's chain is the critical path of the step; a low-priority side stream only gets the compute units the main
# stream's launches leave idle.
SIDE_PRIORITY = {k: {'low': 1, 'normal': 0, 'high': -1}.get(v, 0)
                 for k, v in (kv.split('=') for kv in os.environ.get('MUVO_SIDE_PRIORITY', '').split(',') if '=' in kv)}
_hip_rt = [None]


def _new_stream(device, priority=0):
    """a non-blocking HIP stream of the given priority (-1 high, 0 normal, 1 low) as a torch stream object.  torch.cuda.Stream
    only knows 'normal' and 'high' on ROCm, so other priorities are created through the HIP runtime and wrapped."""
    if priority == 0:
        return torch.cuda.Stream(device=device)
    if _hip_rt[0] is None:
        _hip_rt[0] = C.CDLL('libamdhip64.so')
    rt = _hip_rt[0]
    least, greatest = C.c_int(0), C.c_int(0)
    with torch.cuda.device(device):
        if rt.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) != 0:
            return torch.cuda.Stream(device=device)
        prio = least.value if priority > 0 else greatest.value
        h = C.c_void_p(0)
        if rt.hipStreamCreateWithPriority(C.byref(h), C.c_uint(1), C.c_int(prio)) != 0 or not h.value:   # 1 = hipStreamNonBlocking
            return torch.cuda.Stream(device=device)
    return torch.cuda.ExternalStream(h.value, device=device)


def mark_inputs_ready(device):
    """Everything queued so far on the current stream (packed weights, preprocessed batch) is what a branch's first kernels
    depend on: a side stream waits for THIS point, not for the main stream's later work."""
    if STREAMS and torch.device(device).type == 'cuda':
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        _inputs_ready[torch.device(device).index] = ev


def join_side_streams(device=None, into=None):
    """Make `into` (default: the current stream) wait for everything queued on the side streams."""
    _join_queued[0] = False        # (a backward pass that raised never ran its callback)
    if not _side_streams:
        return
    for (name, idx), st in _side_streams.items():
        if device is not None and torch.device(device).index not in (None, idx):
            continue
        (into if into is not None else torch.cuda.current_stream(st.device)).wait_stream(st)


BRANCHES = set(os.environ.get('MUVO_STREAM_BRANCHES', 'route,lidar,decoders,wgrad').split(','))


def stream_event(device):
    """An event at the current point of the current stream (None when side streams are off): `branch(after=...)`."""
    if not (STREAMS and torch.device(device).type == 'cuda'):
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    return ev


class branch:
    """br = ops.branch('route', 'route', dev, inputs=(x,)); with br: y = br.out(f(x)); ...; br.join() - runs f's kernels on
    the side stream `name` (if the branch group `group` is enabled).  The side stream starts after the event `after`
    (default: the last mark_inputs_ready(); neither: after everything queued on the current stream so far); the current
    stream waits for the branch in join() - call it right before the first consumer of the outputs.  Disabled
    (MUVO_STREAMS=0, group not in MUVO_STREAM_BRANCHES, CPU tensors): a no-op."""

    def __init__(self, group, name, device, inputs=(), after=None):
        device = torch.device(device)
        self.on = STREAMS and group in BRANCHES and device.type == 'cuda'
        self.device, self.name, self.inputs, self.after = device, name, inputs, after
        self.outs, self.joined = [], False

    def __enter__(self):
        if not self.on:
            return self
        self.main = torch.cuda.current_stream(self.device)
        self.side = side_stream(self.name, self.device)
        if self.side == self.main:
            self.on = False
            return self
        ev = self.after if self.after is not None else _inputs_ready.get(self.device.index)
        if ev is None:
            self.side.wait_stream(self.main)
        else:
            self.side.wait_event(ev)
        for t in self.inputs:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def out(self, t):
        """declare a tensor (list / dict of tensors) that leaves the branch: used by the main stream after join()"""
        if self.on:
            for x in (t.values() if isinstance(t, dict) else t if isinstance(t, (list, tuple)) else (t,)):
                if torch.is_tensor(x) and x.is_cuda:
                    self.outs.append(x)
        return t

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on and not self.joined:
            self.joined = True
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.side)
            for x in self.outs:
                x.record_stream(cur)


# Weight gradients on their own stream.  In the backward pass of a convolution only the data gradient is on the critical path
# (the next layer's backward needs it); the weight gradient is needed by the optimizer.  Issued on the stream 'wgrad' it runs
# next to the following layers' data-gradient kernels and fills the compute units their tails and small launches leave idle.
# What the two streams share: the split planes of dy - written by the main stream (prepare_dy / dgrad), read by the weight
# gradient - come from a ring of WGRAD_RING buffers, a buffer is rewritten only after the weight gradient that read it has
# finished (event); tensors owned by autograd (x, y, dy, kept planes of x) are marked with record_stream.
WGRAD_STREAM = STREAMS and 'wgrad' in set(os.environ.get('MUVO_STREAM_BRANCHES', 'route,lidar,decoders,wgrad').split(','))
WGRAD_RING = int(os.environ.get('MUVO_WGRAD_RING', '8'))      # same-box A/B, ms/step: 3 slots 86.8, 6: 86.4, 8: 85.95, 12: 85.9 (+8.6 GB of HBM at 8)
_dy_rings = {}


class _DySlot:
    __slots__ = ('t', 'done')

    def __init__(self):
        self.t, self.done = None, None


def _dy_ws_acquire(nfloats, device):
    """next buffer of the current stream's ring of dy-plane workspaces (waits for the weight gradient that last read it)"""
    key = (torch.device(device).index, _stream_key(device))
    ring = _dy_rings.get(key)
    if ring is None:
        ring = _dy_rings[key] = dict(i=0, slots=[_DySlot() for _ in range(WGRAD_RING)])
    ring['i'] = (ring['i'] + 1) % WGRAD_RING
    sl = ring['slots'][ring['i']]
    if sl.done is not None:
        torch.cuda.current_stream(device).wait_event(sl.done)
        sl.done = None
    if sl.t is None or sl.t.numel() < nfloats:
        sl.t = torch.empty(int(nfloats), device=device, dtype=torch.float32)
    return sl


def wgrad_stream(device):
    """the side stream weight gradients are issued on, or None (switched off / CPU / already on it)"""
    if not WGRAD_STREAM or torch.device(device).type != 'cuda':
        return None
    return side_stream('wgrad', device)


# ------------------------------------------------------------------------------------------------
# optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline object)
KERNEL_TIMING = None


class _NoEvent:
    def record(self):
        pass


_NO_EVENT = _NoEvent()


class KernelTiming:
    """only: record events for this kernel class alone (bench.py brackets just the dominant class inside the timed region:
    ~1400 event pairs per step around every conv call cost 2.4 ms/step, ~140 pairs cost 0.2)."""

    def __init__(self, only=None, prealloc=0):
        """prealloc: events created AND recorded once up front.  A torch.cuda.Event gets its HIP event at its first record(), and
        the first few thousand creations of a process grow the runtime's signal pools: bench.py's timed region paid ~100 ms for
        that in its first three steps (105 / 107 / 125 ms against 77) until the pool was filled ahead of it."""
        self.rec = []
        self.only = only
        self.enabled = True          # bench.py brackets every fourth timed step only (each pair costs ~1.5 us of host and queue time)
        self.pool = []
        if prealloc > 0:
            self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(int(prealloc))]
            for e in self.pool:
                e.record()
            torch.cuda.synchronize()

    def bracket(self, cls, flops, launches, tag='', nbytes=0.0):
        if not self.enabled or (self.only is not None and cls.split(':')[0] != self.only):
            return _NO_EVENT, _NO_EVENT
        e0 = self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)
        e1 = self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)
        self.rec.append((cls, flops, launches, e0, e1, tag, nbytes))
        return e0, e1

    def layer_table(self):
        """Per (kernel class, layer shape) rows sorted by time: the optimisation worklist."""
        torch.cuda.synchronize()
        agg = {}
        for cls, flops, launches, e0, e1, tag, _nb in self.rec:
            a = agg.setdefault((cls, tag), [0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += flops
            a[2] += 1
        rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
        out = [f'{"ms/call":>9} {"calls":>5} {"TFLOP/s":>8} {"GFLOP":>9}  class | layer']
        for (cls, tag), (sec, fl, n) in rows:
            out.append(f'{sec / n * 1e3:9.3f} {n:5d} {fl / max(sec, 1e-12) / 1e12:8.1f} {fl / n / 1e9:9.2f}  {cls} | {tag}')
        return '\n'.join(out)

    def summary(self):
        """(roofline object of the dominant conv kernel family, per-class table).  Times are HIP-event brackets on the
        launch stream around the C-ABI call (they include the small pack/split/unpack helper launches of that call)."""
        torch.cuda.synchronize()
        agg = {}
        for cls, flops, launches, e0, e1, _tag, nb in self.rec:
            a = agg.setdefault(cls, [0.0, 0.0, 0, 0.0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += flops
            a[2] += launches
            a[3] += nb
        classes = {k: dict(seconds=v[0], tflop=v[1] / 1e12, launches=v[2], algorithmic_gb=v[3] / 1e9,
                           avg_launch_us=v[0] / max(v[2], 1) * 1e6, tflops=v[1] / max(v[0], 1e-12) / 1e12)
                   for k, v in agg.items()}
        fam = {}
        for k, c in classes.items():
            f = fam.setdefault(k.split(':')[0], dict(seconds=0.0, tflop=0.0, launches=0, gb=0.0))
            f['seconds'] += c['seconds']
            f['tflop'] += c['tflop']
            f['launches'] += c['launches']
            f['gb'] += c['algorithmic_gb']
        mfma = {k: v for k, v in fam.items() if k in PEAK_TFLOPS}
        roof = None
        if mfma:
            dom = max(mfma, key=lambda k: mfma[k]['seconds'])
            f = mfma[dom]
            alg = f['tflop'] / max(f['seconds'], 1e-12)
            mult = MFMA_FLOPS_PER_ALGORITHMIC_FLOP[dom]
            if dom in ('bf16x3_implicit_gemm', 'bf16x3_small_tile') and lib().muvo_conv_get_products() == 1:
                mult = 1.0        # CONV_BF16: one MFMA product per algorithmic product
            roof = dict(bound='mfma', kernel=KERNEL_NAMES[dom], achieved=alg * mult, peak=PEAK_TFLOPS[dom], unit='TFLOP/s',
                        frac=alg * mult / PEAK_TFLOPS[dom], traffic=None, algorithmic_tflops=alg,
                        mfma_flops_per_algorithmic_flop=mult, launches=f['launches'],
                        algorithmic_bytes=f['gb'] * 1e9 / max(f['launches'], 1),
                        algorithmic_bytes_unit='bytes per launch: fp32 operand tensors read once + result written once '
                                               '(4 * (N*Cin*in_pixels + N*Cout*out_pixels + weight elements))',
                        avg_launch_us=f['seconds'] / max(f['launches'], 1) * 1e6,
                        share_of_conv_time=f['seconds'] / max(sum(v['seconds'] for v in fam.values()), 1e-12))
        return roof, classes


FAMILY = {0: 'f32_implicit_gemm', 1: 'bf16x3_implicit_gemm', 2: 'vox_4x4x1', 3: 'heads_valu', 4: 'vox_bf16x3',
          5: 'bf16x3_small_tile', -1: 'unknown'}
KERNEL_NAMES = {'f32_implicit_gemm': 'conv_fwd_kernel / conv_wgrad_kernel (v_mfma_f32_32x32x2_f32)',
                'bf16x3_implicit_gemm': 'conv_bf3_kernel<256,128|128,256> / conv_bf3_wgrad_pp_kernel: eight-wave ping-pong tiles '
                                        '(v_mfma_f32_32x32x16_bf16, 3 products)',
                'bf16x3_small_tile': 'conv_bf3_kernel<64,128> / conv_bf3_wgrad_kernel: four-wave tiles of the same arithmetic',
                'vox_4x4x1': 'vox_conv_kernel / vox_wgrad_kernel (v_mfma_f32_4x4x1_16b_f32)',
                'vox_bf16x3': 'vox_bf3_kernel (v_mfma_f32_16x16x32_bf16, 3 products, rows padded to 16)'}
# dense MFMA peaks from /opt/skills/guides/MI355X_MICROARCH.md
PEAK_TFLOPS = {'f32_implicit_gemm': 157.3, 'bf16x3_implicit_gemm': 2500.0, 'bf16x3_small_tile': 2500.0, 'vox_4x4x1': 157.3,
               'vox_bf16x3': 2500.0}
MFMA_FLOPS_PER_ALGORITHMIC_FLOP = {'f32_implicit_gemm': 1.0, 'bf16x3_implicit_gemm': 3.0, 'bf16x3_small_tile': 3.0,
                                   'vox_4x4x1': 1.0, 'vox_bf16x3': 3.0}


def _conv_tag(geom, n, in_sz):
    return (f'{"convT" if geom.transposed else "conv"}{geom.nd}d {geom.cin}->{geom.cout} k{geom.ksz} s{geom.stride} '
            f'n{n} in{tuple(in_sz)}')


def _conv_bytes(geom, n, in_sz, out_sz):
    """Algorithmic HBM bytes of one conv operation (forward, data gradient or weight gradient alike): both activation
    tensors and the weight, fp32, each touched once."""
    import math
    return 4.0 * (n * geom.cin * math.prod(in_sz) + n * geom.cout * math.prod(out_sz)
                  + geom.cin * geom.cout * math.prod(geom.ksz))


def _conv_flops(geom, n, in_sz, out_sz):
    import math
    taps = math.prod(geom.ksz)
    pix = math.prod(in_sz) if geom.transposed else math.prod(out_sz)
    return 2.0 * n * geom.cin * geom.cout * taps * pix


# ================================================================================================ GEMM
def gemm(A, B, Cout, M, N, K, sam, sak, sbk, sbn, scm, bias=None, alpha=1.0, act=ACT_NONE, slope=0.0, mode=0,
         B1=1, B2=1, a_b=(0, 0), b_b=(0, 0), c_b=(0, 0), bias_div=1, a_off=0, b_off=0, c_off=0):
    """Raw strided GEMM on storage pointers (+ element offsets)."""
    d = GemmDesc(M, N, K, sam, sak, sbk, sbn, scm, B1, B2, a_b[0], a_b[1], b_b[0], b_b[1], c_b[0], c_b[1], alpha,
                 bias_div, act, slope, mode)
    pa = C.c_void_p(A.data_ptr() + 4 * a_off)
    pb = C.c_void_p(B.data_ptr() + 4 * b_off)
    pc = C.c_void_p(Cout.data_ptr() + 4 * c_off)
    _ck(lib().muvo_gemm(C.byref(d), pa, pb, pc, _f(bias), _st()))


class LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) on the last dim; x (..., in) contiguous.  W grads go to W.grad."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        x = x.contiguous()
        out_f, in_f = weight.shape
        rows = x.numel() // in_f
        y = torch.empty(*x.shape[:-1], out_f, device=x.device, dtype=torch.float32)
        gemm(x, weight, y, rows, out_f, in_f, in_f, 1, 1, in_f, out_f, bias=bias, act=act, slope=slope)
        ctx.act, ctx.slope = act, slope
        ctx.weight, ctx.bias = weight, bias
        ctx.save_for_backward(x, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        out_f, in_f = weight.shape
        rows = x.numel() // in_f
        dy = dy.contiguous()
        if ctx.act != ACT_NONE:
            dz = torch.empty_like(dy)
            _ck(lib().muvo_act_bwd(_f(y), _f(dy), _f(dz), _i64(dy.numel()), ctx.act, _fl(ctx.slope), _st()))
        else:
            dz = dy
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            # dx[r][i] = sum_o dz[r][o] W[o][i]
            gemm(dz, weight, dx, rows, in_f, out_f, out_f, 1, in_f, 1, in_f)
        if weight.requires_grad:
            # dW[o][i] += sum_r dz[r][o] x[r][i]
            gemm(dz, x, grad_of(weight), out_f, in_f, rows, 1, out_f, in_f, 1, in_f, mode=1)
            if bias is not None:
                _ck(lib().muvo_colsum_acc(_f(dz), _f(grad_of(bias)), _i64(rows), _i64(out_f), _i64(out_f), _st()))
        return dx, None, None, None, None


GROUPED_LINEAR = os.environ.get('MUVO_GROUPED_LINEAR', '1') != '0' and os.environ.get('MUVO_DETERMINISTIC', '0') in ('', '0')


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


class GroupedLinearFn(torch.autograd.Function):
    """(y_1, ..., y_L) = (x W_1^T + b_1, ..., x W_L^T + b_L) for L Linear layers that read the same few-row input: the AdaIN
    style projections of a decoder in one launch per pass (csrc/gemm.hip: grouped_linear_*).  args: x, then W_1, b_1, ..."""

    @staticmethod
    def forward(ctx, x, *wb):
        x = x.contiguous()
        ws, bs = list(wb[0::2]), list(wb[1::2])
        m, k = x.shape
        ns = [w.shape[0] for w in ws]
        ys = [torch.empty(m, n, device=x.device, dtype=torch.float32) for n in ns]
        narr = (C.c_int * len(ns))(*ns)
        _ck(lib().muvo_grouped_linear_fwd(_f(x), m, k, len(ws), _ptr_array(ws), _ptr_array(bs), _ptr_array(ys), narr, _st()))
        ctx.ws, ctx.bs, ctx.ns = ws, bs, ns
        ctx.save_for_backward(x)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        x, = ctx.saved_tensors
        ws, bs, ns = ctx.ws, ctx.bs, ctx.ns
        m, k = x.shape
        dys = [torch.zeros(m, n, device=x.device, dtype=torch.float32) if d is None else d.contiguous() for d, n in zip(dys, ns)]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dws = [grad_of(w) if w.requires_grad else None for w in ws]
        dbs = [grad_of(b) if (b is not None and b.requires_grad) else None for b in bs]
        narr = (C.c_int * len(ns))(*ns)
        _ck(lib().muvo_grouped_linear_bwd(_f(x), m, k, len(ws), _ptr_array(ws), _ptr_array(dys), _f(dx) if dx is not None else None,
                                          _ptr_array(dws), _ptr_array(dbs), narr, _st()))
        return (dx,) + (None,) * (2 * len(ws))


def grouped_linear_supported(x, linears):
    if not GROUPED_LINEAR or x.dim() != 2 or not (1 <= len(linears) <= 16):
        return False
    m, k = x.shape
    if m > 24 or k < 64 or k % 4:
        return False
    return all(l.weight.shape[1] == k and l.weight.is_contiguous() and l.weight.data_ptr() % 16 == 0 and
               (l.weight.grad is None or l.weight.grad.data_ptr() % 16 == 0) and
               getattr(l.weight, '_muvo_flat_grad', l.weight).data_ptr() % 16 == 0 for l in linears)


def grouped_linear(x, linears):
    """styles of `linears` (nn.Linear-like modules with .weight / .bias) applied to the same 2-D input, as a tuple"""
    args = []
    for l in linears:
        args += [l.weight, l.bias]
    return GroupedLinearFn.apply(x, *args)


class _LinearPacked:
    __slots__ = ('fwd', 'dgr', 'fwd_key', 'dgr_key', '__weakref__')

    def __init__(self):
        self.fwd = self.dgr = self.fwd_key = self.dgr_key = None


class LinearBf16x3Fn(torch.autograd.Function):
    """The same Linear for token matrices with thousands of rows (transformer encoder) on the bf16x3 implicit-GEMM
    kernels (include/muvo_hip.h: muvo_linear_bf16x3_*): x and dz are split once into bf16 hi/lo planes; forward, data
    gradient and weight gradient all read those planes."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        x = x.contiguous()
        out_f, in_f = weight.shape
        rows = x.numel() // in_f
        L = lib()
        pk = getattr(weight, '_bf3_packed', None)
        if pk is None:
            pk = weight._bf3_packed = _LinearPacked()
        ff, df = C.c_int64(0), C.c_int64(0)
        _ck(L.muvo_linear_bf16x3_pack_floats(in_f, out_f, C.byref(ff), C.byref(df)))
        k = _wkey(weight)
        if pk.fwd is None or pk.fwd_key != k:
            if pk.fwd is None:
                pk.fwd = torch.empty(ff.value, device=x.device, dtype=torch.float32)
            _ck(L.muvo_linear_bf16x3_pack(in_f, out_f, _f(weight), _f(pk.fwd), None, _st()))
            pk.fwd_key = k
            _PACKS.register((id(pk), _plan_epoch[0]), ('lin', (in_f, out_f), weight, weakref.ref(pk), _plan_epoch[0]))
        keep = ctx.needs_input_grad[1]      # a backward follows: wgrad reuses the planes of x
        nws = (L.muvo_linear_bf16x3_workspace_bytes(_i64(rows), in_f) + 3) // 4
        ws_x = torch.empty(nws, device=x.device, dtype=torch.float32) if keep else scratch('lin_ws_x', nws, x.device)
        _ck(L.muvo_linear_bf16x3_split(_f(x), _i64(rows), in_f, _p(ws_x), _st()))
        y = torch.empty(*x.shape[:-1], out_f, device=x.device, dtype=torch.float32)
        _ck(L.muvo_linear_bf16x3_forward(_i64(rows), in_f, out_f, _p(ws_x), _f(pk.fwd), _f(bias), _f(y), act, _fl(slope),
                                         _st()))
        ctx.act, ctx.slope, ctx.weight, ctx.bias, ctx.pk = act, slope, weight, bias, pk
        ctx.ws_x = ws_x if keep else None
        ctx.dims = (rows, in_f, out_f, df.value)
        ctx.x_shape = x.shape
        ctx.save_for_backward(y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        weight, bias, pk = ctx.weight, ctx.bias, ctx.pk
        rows, in_f, out_f, df = ctx.dims
        L = lib()
        dy = dy.contiguous()
        if ctx.act != ACT_NONE:
            dz = torch.empty_like(dy)
            _ck(L.muvo_act_bwd(_f(y), _f(dy), _f(dz), _i64(dy.numel()), ctx.act, _fl(ctx.slope), _st()))
        else:
            dz = dy
        nws = (L.muvo_linear_bf16x3_workspace_bytes(_i64(rows), out_f) + 3) // 4
        ws_dz = scratch('lin_ws_dz', nws, dy.device)
        _ck(L.muvo_linear_bf16x3_split(_f(dz), _i64(rows), out_f, _p(ws_dz), _st()))
        dx = None
        if ctx.needs_input_grad[0]:
            k = _wkey(weight)
            if pk.dgr is None or pk.dgr_key != k:
                if pk.dgr is None:
                    pk.dgr = torch.empty(df, device=dy.device, dtype=torch.float32)
                _ck(L.muvo_linear_bf16x3_pack(in_f, out_f, _f(weight), None, _f(pk.dgr), _st()))
                pk.dgr_key = k
            dx = torch.empty(ctx.x_shape, device=dy.device, dtype=torch.float32)
            _ck(L.muvo_linear_bf16x3_dgrad(_i64(rows), in_f, out_f, _p(ws_dz), _f(pk.dgr), _f(dx), _st()))
        if weight.requires_grad:
            sc = scratch_zeroed('lin_wgrad', out_f * in_f, dy.device)
            _ck(L.muvo_linear_bf16x3_wgrad(_i64(rows), in_f, out_f, _p(ctx.ws_x), _p(ws_dz), _f(sc), _f(grad_of(weight)),
                                           _st()))
            if bias is not None:
                _ck(L.muvo_colsum_acc(_f(dz), _f(grad_of(bias)), _i64(rows), _i64(out_f), _i64(out_f), _st()))
        return dx, None, None, None, None


LINEAR_BF16X3_MIN_ROWS = int(os.environ.get('MUVO_LINEAR_BF16X3_MIN_ROWS', '2048'))


def linear(x, weight, bias=None, act=ACT_NONE, slope=0.0):
    out_f, in_f = weight.shape
    if (x.numel() // in_f >= LINEAR_BF16X3_MIN_ROWS and in_f % 8 == 0 and out_f % 16 == 0 and in_f >= 64 and out_f >= 64
            and get_conv_mode() == CONV_BF16X3):
        return LinearBf16x3Fn.apply(x, weight, bias, act, slope)
    return LinearFn.apply(x, weight, bias, act, slope)


# ================================================================================================ conv
class ConvGeom:
    """Static geometry of a conv layer (everything but the batch/input size)."""

    def __init__(self, nd, transposed, cin, cout, ksz, stride=1, pad=0, dil=1, out_pad=0):
        def t3(v):
            v = (v,) * nd if isinstance(v, int) else tuple(v)
            return (1,) * (3 - nd) + v if nd == 2 and len(v) == 2 else v
        self.nd, self.transposed, self.cin, self.cout = nd, int(transposed), cin, cout
        self.ksz, self.stride, self.dil = t3(ksz), t3(stride), t3(dil)
        p, op = t3(pad), t3(out_pad)
        if nd == 2:
            p = (0,) + p[1:]
            op = (0,) + op[1:]
            self.stride = (1,) + self.stride[1:]
            self.dil = (1,) + self.dil[1:]
        self.pad, self.out_pad = p, op
        self._plans = {}
        self.ws_bytes = {}
        self.family = {}
        self.tclass = {}

    def out_size(self, in_sz):
        o = []
        for a in range(3):
            if not self.transposed:
                o.append((in_sz[a] + 2 * self.pad[a] - self.dil[a] * (self.ksz[a] - 1) - 1) // self.stride[a] + 1)
            else:
                o.append((in_sz[a] - 1) * self.stride[a] - 2 * self.pad[a] + self.dil[a] * (self.ksz[a] - 1) + 1 +
                         self.out_pad[a])
        return tuple(o)

    def plan(self, n, in_sz):
        key = (n, in_sz, _plan_epoch[0])
        pl = self._plans.get(key)
        if pl is None:
            out_sz = self.out_size(in_sz)
            d = ConvDesc(self.nd, self.transposed, n, self.cin, self.cout, (C.c_int32 * 3)(*in_sz),
                         (C.c_int32 * 3)(*out_sz), (C.c_int32 * 3)(*self.ksz), (C.c_int32 * 3)(*self.stride),
                         (C.c_int32 * 3)(*self.pad), (C.c_int32 * 3)(*self.dil))
            ff, df = C.c_int64(0), C.c_int64(0)
            _ck(lib().muvo_conv_pack_sizes(C.byref(d), C.byref(ff), C.byref(df)))
            wsf, wsd, wsx, wsy = (lib().muvo_conv_workspace_bytes(C.byref(d), op) for op in (0, 1, 2, 3))
            if min(wsf, wsd, wsx, wsy) < 0:
                raise RuntimeError(f'muvo_hip error: {lib().muvo_last_error().decode()}')
            self.ws_bytes[(n, in_sz, _plan_epoch[0])] = (wsf, wsd, wsx, wsy)
            self.family[(n, in_sz, _plan_epoch[0])] = tuple(lib().muvo_conv_kernel_family(C.byref(d), op) for op in (0, 1, 2))
            # timing label: family 1 launches on the four-wave small tiles are reported as their own class (5)
            self.tclass[(n, in_sz, _plan_epoch[0])] = tuple(
                5 if (f == 1 and lib().muvo_conv_kernel_variant(C.byref(d), op) == 0) else f
                for op, f in enumerate(self.family[(n, in_sz, _plan_epoch[0])]))
            pl = (d, out_sz, ff.value, df.value)
            self._plans[key] = pl
        return pl


class _PackedWeights:
    """Per-layer K-major packed copies of the weight, refreshed when the weight changed."""

    def __init__(self):
        self.fwd = self.dgr = None
        self.fwd_key = self.dgr_key = None
        self.fwd_plan = self.dgr_plan = None     # (input size, plan epoch) the copy was packed for (layout is batch-independent)


def _wkey(w):
    return (w._version, _weight_epoch[0], w.data_ptr())


def _head_alias(geom, key, op, weight):
    """The 1x1 head kernels (muvo_conv_kernel_family == 3, conv_pw.hip) read the weight in PyTorch's layout: their "packed copy" is
    the parameter itself - no device-to-device copy per head, direction and optimizer step (18 launches per step at base_1d)."""
    fam = geom.family.get(key)
    return fam is not None and fam[op] == 3 and weight.is_contiguous()


def _is_alias(buf, weight):
    return buf is not None and buf.data_ptr() == weight.data_ptr()


# ------------------------------------------------------------------------------------------------
# Batched weight packing (include/muvo_hip.h: muvo_pack_table_*).  Layers register themselves the first time they pack;
# repack_all() (called once per training forward) refreshes every registered copy that already exists with ONE launch and
# marks it current, so the per-layer staleness checks in ConvFn / LinearBf16x3Fn find nothing to do.
class _PackRegistry:
    def __init__(self):
        self.entries, self.seen = [], set()
        self.sig, self.dev, self.n, self.nblk, self.batched = None, None, 0, 0, []

    def register(self, key, entry):
        """entry: (kind, desc, weight, weakref to the layer's packed-copy holder, plan key)"""
        if key not in self.seen:
            self.seen.add(key)
            self.entries.append(entry)


_PACKS = _PackRegistry()
_PACK_BATCH = os.environ.get('MUVO_PACK_BATCH', '1') != '0'


def repack_all():
    flush_bn_counters()
    R = _PACKS
    if not R.entries or not _PACK_BATCH:
        return
    L = lib()
    ptr = lambda t: 0 if t is None else t.data_ptr()
    # drop layers that no longer exist and copies that belong to an earlier plan epoch (conv mode / policy change: their
    # buffers are re-sized lazily by the layer itself, which then registers again)
    live = []
    for e in R.entries:
        pk = e[3]()
        if pk is None or (e[0] == 'conv' and e[4][1] != _plan_epoch[0]) or (e[0] == 'lin' and e[4] != _plan_epoch[0]):
            R.seen.discard((id(pk) if pk is not None else None, e[4]))
            continue
        live.append(e)
    if len(live) != len(R.entries):
        R.entries = live
        R.seen = {(id(e[3]()), e[4]) for e in live}
        R.sig = None
    if not R.entries:
        return
    ents = [(e[0], e[1], e[2], e[3](), e[4]) for e in R.entries]
    sig = tuple((ptr(e[3].fwd), ptr(e[3].dgr), e[2].data_ptr()) for e in ents)
    if sig != R.sig:
        item = L.muvo_pack_table_item_bytes()
        cap = 16 * len(R.entries) + 8
        host = torch.zeros(cap * item, dtype=torch.uint8)
        n, nblk = C.c_int(0), C.c_int64(0)
        R.batched = []
        for e, orig in zip(ents, R.entries):
            kind, desc, weight, pk, _ = e
            if pk.fwd is None and pk.dgr is None:
                continue
            if _is_alias(pk.fwd, weight) or _is_alias(pk.dgr, weight):      # 1x1 heads: the kernels read the parameter itself
                continue
            if kind == 'conv':
                rc = L.muvo_conv_pack_table_add(C.c_void_p(host.data_ptr()), cap, C.byref(n), C.byref(nblk), C.byref(desc),
                                                _f(weight), _f(pk.fwd), _f(pk.dgr))
            else:
                rc = L.muvo_linear_bf16x3_pack_table_add(C.c_void_p(host.data_ptr()), cap, C.byref(n), C.byref(nblk),
                                                         desc[0], desc[1], _f(weight), _f(pk.fwd), _f(pk.dgr))
            if rc == 0:
                R.batched.append(orig)
            elif rc != 1:
                _ck(rc)
        R.n, R.nblk = n.value, nblk.value
        R.dev = host[:max(R.n, 1) * item].to(ents[0][2].device)
        R.sig = sig
    if R.n:
        _ck(L.muvo_pack_table_run(C.c_void_p(R.dev.data_ptr()), R.n, _i64(R.nblk), _st()))
        for kind, _, weight, pkref, pkey in R.batched:
            pk = pkref()
            if pk is None:
                continue
            k = _wkey(weight)
            if pk.fwd is not None:
                pk.fwd_key = k
            if pk.dgr is not None:
                pk.dgr_key = k
            if kind == 'conv':
                pk.fwd_plan = pkey if pk.fwd is not None else pk.fwd_plan
                pk.dgr_plan = pkey if pk.dgr is not None else pk.dgr_plan


_KEEP_WS = os.environ.get('MUVO_KEEP_WS', '1') != '0'


class ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, geom, packed, act, slope, act_bwd_fused=False, moments=None, aff_src=None, aff=None):
        # aff given: x is the placeholder output of a lazy AdaIN (AdaINFn, lazy=True), the real operand is aff * aff_src + shift,
        # applied by the kernels while they stage aff_src (muvo_conv_forward_affine / muvo_conv_wgrad_affine)
        ctx.aff = aff
        if aff is not None:
            x = aff_src
        # split planes that came with the tensor (written by its producer: BNActFn with planes=True): (workspace, stream key)
        xp = getattr(x, '_muvo_planes', None)
        x_is_placeholder = getattr(x, '_muvo_planes_only', False)
        if not x_is_placeholder:
            x = x.contiguous()
        ctx.act_bwd_fused = act_bwd_fused
        n = x.shape[0]
        in_sz = tuple(x.shape[2:]) if geom.nd == 3 else (1,) + tuple(x.shape[2:])
        d, out_sz, ff, df = geom.plan(n, in_sz)
        L = lib()
        # the 7x7 stride-2 stems of the ResNet-18 trunks: direct convolution on bf16x3 products from the fp32 parameter itself
        # (csrc/conv_stem.hip); only where the arithmetic mode is bf16x3 and the input needs no gradient
        skey = ('stem', n, in_sz, _plan_epoch[0])
        stem = geom.family.get(skey)
        if stem is None:
            stem = geom.family[skey] = bool(L.muvo_stem_conv_supported(C.byref(d)))
        ctx.stem = bool(stem and get_conv_mode() != CONV_F32 and bias is None and act == ACT_NONE and moments is None and aff is None
                        and not x_is_placeholder and not ctx.needs_input_grad[0] and getattr(ctx, '_head_fwd', None) is None
                        and weight.is_contiguous())
        if ctx.stem:
            y = torch.empty((n, geom.cout) + out_sz[1:], device=x.device, dtype=torch.float32)
            kt = KERNEL_TIMING
            if kt is not None:
                e0, e1 = kt.bracket('bf16x3_small_tile:fwd', _conv_flops(geom, n, in_sz, out_sz), 1, _conv_tag(geom, n, in_sz),
                                    _conv_bytes(geom, n, in_sz, out_sz))
                e0.record()
            _ck(L.muvo_stem_conv_forward(C.byref(d), _f(x), _f(weight), None, _f(y), 0, _st()))
            if kt is not None:
                e1.record()
            ctx.geom, ctx.packed, ctx.act, ctx.slope = geom, packed, act, slope
            ctx.weight, ctx.bias, ctx.in_sz, ctx.ws_x, ctx.x_is_placeholder = weight, bias, in_sz, None, False
            ctx.save_for_backward(x, None)
            return y
        k = _wkey(weight)
        pkey = (in_sz, _plan_epoch[0])
        if _head_alias(geom, (n, in_sz, _plan_epoch[0]), 0, weight):
            packed.fwd, packed.fwd_key, packed.fwd_plan = weight.detach().view(-1), k, pkey       # 1x1 heads read the PyTorch layout
        else:
            if packed.fwd is None or packed.fwd.numel() < ff or _is_alias(packed.fwd, weight):
                packed.fwd = torch.empty(ff, device=x.device, dtype=torch.float32)
                packed.fwd_key = None
            if packed.fwd_key != k or packed.fwd_plan != pkey:
                _ck(L.muvo_conv_pack_weights(C.byref(d), _f(weight), _f(packed.fwd), None, _st()))
                packed.fwd_key, packed.fwd_plan = k, pkey
                _PACKS.register((id(packed), pkey), ('conv', d, weight, weakref.ref(packed), pkey))
        oshape = (n, geom.cout) + (out_sz if geom.nd == 3 else out_sz[1:])
        y = torch.empty(oshape, device=x.device, dtype=torch.float32)
        kt = KERNEL_TIMING
        if kt is not None:
            import math
            e0, e1 = kt.bracket(FAMILY[geom.tclass[(n, in_sz, _plan_epoch[0])][0]] + ':fwd', _conv_flops(geom, n, in_sz, out_sz),
                                math.prod(geom.stride) if geom.transposed else 1, _conv_tag(geom, n, in_sz),
                                _conv_bytes(geom, n, in_sz, out_sz))
            e0.record()
        wsb = geom.ws_bytes[(n, in_sz, _plan_epoch[0])]
        hf = getattr(ctx, '_head_fwd', None)
        use_xp = (xp is not None and wsb[0] > 0 and geom.family[(n, in_sz, _plan_epoch[0])][0] == 1 and hf is None and aff is None
                  and moments is None and xp[1] == _stream_key(x.device) and xp[0].numel() * 4 >= wsb[0])
        if x_is_placeholder and not (use_xp and (wsb[2] > 0 or not ctx.needs_input_grad[1])):
            raise RuntimeError('muvo_hip: a planes-only tensor (BNActFn keep_f32=False) reached a convolution that needs its fp32 values')
        # wgrad reuses the split copy of x (grad mode is off inside forward: needs_input_grad says whether a backward follows)
        keep_ws = wsb[0] > 0 and wsb[2] > 0 and ctx.needs_input_grad[1] and _KEEP_WS
        if use_xp:
            ws = xp[0]
            keep_ws = wsb[2] > 0 and ctx.needs_input_grad[1] and (_KEEP_WS or x_is_placeholder)
        elif keep_ws:
            ws = torch.empty((wsb[0] + 3) // 4, device=x.device, dtype=torch.float32)
        else:
            ws = scratch('conv_ws', (wsb[0] + 3) // 4, x.device) if wsb[0] else None
        ctx.ws_x = ws if keep_ws else None
        ctx.x_is_placeholder = x_is_placeholder
        if hf is not None:
            _ck(L.muvo_conv_forward_head(C.byref(d), _f(x), _f(packed.fwd), _f(bias), _f(y), act, _fl(slope), _p(ws),
                                         _f(hf[0]), _f(hf[1]), hf[0].shape[0], _f(hf[2]), _st()))
        elif aff is not None:
            _ck(L.muvo_conv_forward_affine(C.byref(d), _f(x), _f(aff), _f(packed.fwd), _f(bias), _f(y), act, _fl(slope), _p(moments), _st()))
        elif moments is not None:     # instance-norm statistics of y from the epilogue registers (voxel bf16x3 kernels)
            _ck(L.muvo_conv_forward_moments(C.byref(d), _f(x), _f(packed.fwd), _f(bias), _f(y), act, _fl(slope), _p(moments), _st()))
        elif use_xp:
            _ck(L.muvo_conv_forward_planes(C.byref(d), None if x_is_placeholder else _f(x), _f(packed.fwd), _f(bias), _f(y), act,
                                           _fl(slope), _p(ws), 1, _st()))
        else:
            _ck(L.muvo_conv_forward(C.byref(d), _f(x), _f(packed.fwd), _f(bias), _f(y), act, _fl(slope), _p(ws), _st()))
        if kt is not None:
            e1.record()
        ctx.geom, ctx.packed, ctx.act, ctx.slope = geom, packed, act, slope
        ctx.weight, ctx.bias, ctx.in_sz = weight, bias, in_sz
        ctx.save_for_backward(x, y if (act != ACT_NONE and not act_bwd_fused) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        return ConvFn._backward(ctx, dy, None)

    @staticmethod
    def _backward(ctx, dy, head):
        """head: None, or (dlogits, head_weight (CO, Cout)) of a 1x1 output head that reads this layer's output: its data
        gradient W_head^T dlogits is added to dy inside the split pass (muvo_conv_prepare_dy_head) — dy may then be None (the
        head is the only consumer)."""
        x, y = ctx.saved_tensors
        geom, packed, weight, bias = ctx.geom, ctx.packed, ctx.weight, ctx.bias
        d, out_sz, ff, df = geom.plan(x.shape[0], ctx.in_sz)
        L = lib()
        if getattr(ctx, 'stem', False):
            # stem: only the weight gradient exists (the input image needs none); on the weight-gradient stream like the others
            if weight.requires_grad and dy is not None:
                dy = dy.contiguous()
                gw = grad_of(weight)
                cur, wst = torch.cuda.current_stream(x.device), wgrad_stream(x.device)
                kt = KERNEL_TIMING
                side = wst is not None and wst != cur
                if side:
                    wst.wait_stream(cur)
                    x.record_stream(wst)
                    dy.record_stream(wst)
                with (torch.cuda.stream(wst) if side else contextlib.nullcontext()):
                    if kt is not None:
                        e0, e1 = kt.bracket('bf16x3_small_tile:wgrad', _conv_flops(geom, x.shape[0], ctx.in_sz, out_sz), 1,
                                            _conv_tag(geom, x.shape[0], ctx.in_sz), _conv_bytes(geom, x.shape[0], ctx.in_sz, out_sz))
                        e0.record()
                    _ck(L.muvo_stem_conv_wgrad(C.byref(d), _f(x), _f(dy), _f(gw), _st()))
                    if kt is not None:
                        e1.record()
            return None, None, None, None, None, None, None, None, None, None, None
        dyp = getattr(dy, '_muvo_planes', None) if dy is not None else None     # dy arrives as split planes (BNActFn.backward)
        if dy is not None and dyp is None:
            dy = dy.contiguous()
        key = (x.shape[0], ctx.in_sz, _plan_epoch[0])
        fam, wsb = geom.family[key], geom.ws_bytes[key]
        x_ph = getattr(ctx, 'x_is_placeholder', False)
        xptr = ctx.ws_x if x_ph else x            # a planes-only input has no fp32 storage: the kernels read ctx.ws_x (flags bit 0)
        # both gradient kernels read dy through the channels-last split planes: one fused pass makes them from (y, dy)
        # together with the activation derivative and the bias gradient
        fused_dy = (fam[1] == 1 and fam[2] == 1 and ctx.needs_input_grad[0] and weight.requires_grad and wsb[1] > 0
                    and wsb[3] > 0)
        if head is not None:
            gl, head_w, head_b, head_wgrad = head
            co = head_w.shape[0]
            if not (fused_dy and L.muvo_conv_prepare_dy_head_supported(C.byref(d), co)):
                # no fused preamble for this shape: materialise the head's data gradient (dy + W_head^T dlogits)
                hd = torch.einsum('nk...,kc->nc...', gl, head_w.view(co, -1))
                dy = hd.contiguous() if dy is None else dy + hd
                head_wgrad()
                head = None
        ws_dy_fused = None
        wst = wgrad_stream(x.device) if weight.requires_grad else None
        dy_slot = None

        def dy_workspace(nfloats):
            # the dy planes both gradient kernels read: with the weight gradient on its own stream a ring buffer (one per
            # backward call), otherwise the stream's scratch
            nonlocal dy_slot
            if wst is None:
                return scratch('conv_ws_dy', nfloats, x.device)
            if dy_slot is None:
                dy_slot = _dy_ws_acquire(nfloats, x.device)
            assert dy_slot.t.numel() >= nfloats
            return dy_slot.t
        if dyp is not None:
            if not (fused_dy and head is None and bias is None and (ctx.act == ACT_NONE or ctx.act_bwd_fused)
                    and dyp[0].numel() * 4 >= max(wsb[1], wsb[3]) and dyp[1] == _stream_key(x.device)):
                raise RuntimeError('muvo_hip: a gradient that exists only as split planes reached a convolution backward that cannot read them')
            ws_dy_fused = dyp[0]
            dz = dy = dyp[0]          # (pointer stand-in: the bf16x3 kernels read the planes only)
        elif fused_dy:
            ws_dy_fused = dy_workspace((max(wsb[1], wsb[3]) + 3) // 4)
            use_act = ctx.act != ACT_NONE and not ctx.act_bwd_fused
            if head is not None:
                # with an activation the pass reads y anyway: the head's weight / bias gradient rides along
                ride = use_act and head_w.requires_grad
                if not ride:
                    head_wgrad()
                _ck(L.muvo_conv_prepare_dy_head(C.byref(d), _f(y) if use_act else None, _f(dy) if dy is not None else None,
                                                _f(gl.contiguous()), _f(head_w.contiguous().view(co, -1)), co,
                                                ctx.act if use_act else ACT_NONE, _fl(ctx.slope), _p(ws_dy_fused),
                                                _f(grad_of(bias)) if bias is not None else None,
                                                _f(grad_of(head_w)) if ride else None,
                                                _f(grad_of(head_b)) if (ride and head_b is not None) else None, _st()))
                if dy is None:
                    dy = y if y is not None else x     # placeholder pointer: the bf16x3 kernels read the planes only
            else:
                _ck(L.muvo_conv_prepare_dy(C.byref(d), _f(y) if use_act else None, _f(dy), ctx.act if use_act else ACT_NONE,
                                           _fl(ctx.slope), _p(ws_dy_fused), _f(grad_of(bias)) if bias is not None else None,
                                           _st()))
            dz = dy   # not read by the bf16x3 kernels
        elif ctx.act != ACT_NONE and not ctx.act_bwd_fused:
            dz = torch.empty_like(dy)
            _ck(L.muvo_act_bwd(_f(y), _f(dy), _f(dz), _i64(dy.numel()), ctx.act, _fl(ctx.slope), _st()))
        else:
            dz = dy   # no activation, or its derivative was already chained by the consumer's backward
        dx = None
        ws_dy, dy_split = None, False
        if ctx.needs_input_grad[0]:
            k = _wkey(weight)
            if _head_alias(geom, key, 1, weight):
                packed.dgr, packed.dgr_key, packed.dgr_plan = weight.detach().view(-1), k, key[1:]
            else:
                if packed.dgr is None or packed.dgr.numel() < df or _is_alias(packed.dgr, weight):
                    packed.dgr = torch.empty(df, device=x.device, dtype=torch.float32)
                    packed.dgr_key = None
                if packed.dgr_key != k or packed.dgr_plan != key[1:]:
                    _ck(L.muvo_conv_pack_weights(C.byref(d), _f(weight), None, _f(packed.dgr), _st()))
                    packed.dgr_key, packed.dgr_plan = k, key[1:]
            dx = torch.empty(x.shape, device=dz.device, dtype=torch.float32)
            kt = KERNEL_TIMING
            if kt is not None:
                import math
                e0, e1 = kt.bracket(FAMILY[geom.tclass[(x.shape[0], ctx.in_sz, _plan_epoch[0])][1]] + ':dgrad',
                                    _conv_flops(geom, x.shape[0], ctx.in_sz, out_sz),
                                    1 if geom.transposed else math.prod(geom.stride), _conv_tag(geom, x.shape[0], ctx.in_sz),
                                    _conv_bytes(geom, x.shape[0], ctx.in_sz, out_sz))
                e0.record()
            nb = max(wsb[1], wsb[3])
            ws_dy = (ws_dy_fused if dyp is not None else dy_workspace((nb + 3) // 4)) if nb else None
            dy_split = wsb[1] > 0
            _ck(L.muvo_conv_dgrad(C.byref(d), _f(dz), _f(packed.dgr), _f(dx), _p(ws_dy), 1 if fused_dy else 0, _st()))
            if kt is not None:
                e1.record()
        if weight.requires_grad:
            gw = grad_of(weight)
            db = grad_of(bias) if (bias is not None and not fused_dy) else None   # fused_dy: already accumulated
            ws_x = ctx.ws_x
            flags = (1 if ws_x is not None else 0) | (2 if (dy_split and wsb[3] > 0) else 0)
            cur = torch.cuda.current_stream(x.device)
            if wst is not None and wst != cur:
                # everything the weight gradient reads is queued on the current stream by now; it starts behind that point
                wst.wait_stream(cur)
                for t in (xptr, y, dz, ctx.ws_x, ctx.aff):
                    if t is not None:
                        t.record_stream(wst)
                wctx = torch.cuda.stream(wst)
                wctx.__enter__()
            else:
                wctx = None
            try:
                ws = scratch_zeroed('wgrad', ff, x.device)         # (per stream)
                kt = KERNEL_TIMING
                if kt is not None:
                    import math
                    e0, e1 = kt.bracket(FAMILY[geom.tclass[(x.shape[0], ctx.in_sz, _plan_epoch[0])][2]] + ':wgrad',
                                        _conv_flops(geom, x.shape[0], ctx.in_sz, out_sz),
                                        math.prod(geom.stride) if geom.transposed else 1, _conv_tag(geom, x.shape[0], ctx.in_sz),
                                        _conv_bytes(geom, x.shape[0], ctx.in_sz, out_sz))
                    e0.record()
                if ws_x is None and wsb[2]:
                    ws_x = scratch('conv_ws', (wsb[2] + 3) // 4, x.device)
                if wsb[3] and ws_dy is None:
                    ws_dy = scratch('conv_ws_dy', (wsb[3] + 3) // 4, x.device)
                if ctx.aff is not None:
                    _ck(L.muvo_conv_wgrad_affine(C.byref(d), _f(x), _f(ctx.aff), _f(dz), _f(gw), _f(db), _st()))
                else:
                    assert not x_ph or (flags & 1)
                    _ck(L.muvo_conv_wgrad(C.byref(d), _f(xptr), _f(dz), _f(ws), _f(gw), _f(db), _p(ws_x), _p(ws_dy), flags, _st()))
                if kt is not None:
                    e1.record()
                if wctx is not None and dy_slot is not None:
                    dy_slot.done = torch.cuda.Event()
                    dy_slot.done.record(wst)
            finally:
                if wctx is not None:
                    wctx.__exit__(None, None, None)
        return dx, None, None, None, None, None, None, None, None, None, None


def conv(x, weight, bias, geom, packed, act=ACT_NONE, slope=0.0, act_bwd_fused=False, moments=None, lazy=None):
    """lazy: (raw, aff) of a lazy AdaIN whose placeholder output is x (adain_lazy); the convolution applies the AdaIN itself."""
    if lazy is not None:
        return ConvFn.apply(x, weight, bias, geom, packed, act, slope, act_bwd_fused, moments, lazy[0], lazy[1])
    y = ConvFn.apply(x, weight, bias, geom, packed, act, slope, act_bwd_fused, moments)
    if (BN_PLANES and bias is None and act == ACT_NONE and moments is None and x.requires_grad and weight.requires_grad
            and torch.is_grad_enabled()):
        # the backward of this convolution reads dy as split planes (ConvFn._backward: fused_dy): a BatchNorm behind it may hand
        # its dx over in that form (bn_act(from_conv=True))
        n = x.shape[0]
        in_sz = tuple(x.shape[2:]) if geom.nd == 3 else (1,) + tuple(x.shape[2:])
        key = (n, in_sz, _plan_epoch[0])
        fam, wsb = geom.family.get(key), geom.ws_bytes.get(key)
        if fam is not None and fam[1] == 1 and fam[2] == 1 and wsb[1] > 0 and wsb[3] > 0:
            y._muvo_dy_planes_ok = True
    return y


CONV_AFFINE = os.environ.get('MUVO_CONV_AFFINE', '1') != '0'


def conv_affine_supported(x, geom, moments):
    """can the convolution `geom` consume x = the (not yet normalised) output of the previous layer together with that layer's
    AdaIN as a per-(n, channel) scale / shift (muvo_conv_forward_affine)?  Needs the statistics from the producer's epilogue."""
    if not CONV_AFFINE or moments is None or x.dim() != 5 or not (x.requires_grad and torch.is_grad_enabled()):
        return False
    n = x.shape[0]
    in_sz = tuple(x.shape[2:])
    d = geom.plan(n, in_sz)[0]
    key = ('affine', n, in_sz, _plan_epoch[0])
    ok = geom.family.get(key)
    if ok is None:
        ok = geom.family[key] = bool(lib().muvo_conv_affine_supported(C.byref(d)))
    return ok


CONV_MOMENTS = os.environ.get('MUVO_CONV_MOMENTS', '1') != '0'


def conv_moments_buffer(x, geom):
    """A zeroed (N, Cout, 2) float64 buffer if the convolution `geom` on input x can deliver the instance-norm statistics of
    its output from its epilogue (muvo_conv_forward_moments), else None.  The buffer is per layer and stays all-zero between
    uses (muvo_adain_fwd_moments clears it)."""
    if not CONV_MOMENTS:
        return None
    n = x.shape[0]
    in_sz = tuple(x.shape[2:]) if geom.nd == 3 else (1,) + tuple(x.shape[2:])
    d = geom.plan(n, in_sz)[0]
    key = ('moments', n, in_sz, _plan_epoch[0])
    ok = geom.family.get(key)
    if ok is None:
        ok = geom.family[key] = bool(lib().muvo_conv_forward_moments_supported(C.byref(d)))
    if not ok:
        return None
    cache = geom.__dict__.setdefault('_moments', {})       # one zeroed buffer per batch size (training / validation / imagination)
    buf = cache.get((n, x.device))
    if buf is None:
        buf = cache[(n, x.device)] = torch.zeros(n, geom.cout, 2, device=x.device, dtype=torch.float64)
        _MOMENT_BUFS.append(weakref.ref(buf))
    return buf


def _grad_is_private(g):
    """May a backward function write into the gradient tensor it was handed?  Only if nobody else holds it: autograd hands the SAME
    tensor to several consumers when a gradient is shared (both inputs of an add, expand views, retained grads / hooks).  Inside
    backward() a tensor made for this one consumer has two owners (the engine's input buffer and the Python argument); a shared
    one has more, or is a view (tests/test_kernels_gpu.py::test_head_branch_shared_gradient)."""
    return g._base is None and g._use_count() <= 2


class HeadBranchFn(torch.autograd.Function):
    """x -> (x, head(x)) for a 1x1 output head that hangs off a decoder trunk (ConvDecoder / VoxelDecoder1: head_4 and head_2
    read the feature map that also feeds the next stage, common.py:520-545,608-632).  Forward is the plain head convolution.
    Backward: the trunk's gradient arrives as the gradient of the first output; the head's data gradient is ACCUMULATED into
    that tensor by the kernel (muvo_conv_dgrad_accumulate) instead of autograd adding two full feature maps in a separate
    pass (3 x 681 MB at the rgb 1/2-scale level)."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, packed):
        y = ConvFn.forward(ctx, x, weight, bias, geom, packed, ACT_NONE, 0.0)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, gx, gy):
        x, _ = ctx.saved_tensors
        geom, packed, weight, bias = ctx.geom, ctx.packed, ctx.weight, ctx.bias
        if gy is None:
            return gx, None, None, None, None
        d, out_sz, ff, df = geom.plan(x.shape[0], ctx.in_sz)
        L = lib()
        gy = gy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            k, pkey = _wkey(weight), (ctx.in_sz, _plan_epoch[0])
            if _head_alias(geom, (x.shape[0], ctx.in_sz, _plan_epoch[0]), 1, weight):
                packed.dgr, packed.dgr_key, packed.dgr_plan = weight.detach().view(-1), k, pkey
            else:
                if packed.dgr is None or packed.dgr.numel() < df or _is_alias(packed.dgr, weight):
                    packed.dgr = torch.empty(df, device=x.device, dtype=torch.float32)
                    packed.dgr_key = None
                if packed.dgr_key != k or packed.dgr_plan != pkey:
                    _ck(L.muvo_conv_pack_weights(C.byref(d), _f(weight), None, _f(packed.dgr), _st()))
                    packed.dgr_key, packed.dgr_plan = k, pkey
            if gx is not None and gx.is_contiguous() and _grad_is_private(gx):
                dx = gx            # a fresh tensor made by the trunk's backward for this one consumer: accumulated in place
                _ck(L.muvo_conv_dgrad_accumulate(C.byref(d), _f(gy), _f(packed.dgr), _f(dx), _st()))
            else:
                dx = torch.empty_like(x)
                _ck(L.muvo_conv_dgrad(C.byref(d), _f(gy), _f(packed.dgr), _f(dx), None, 0, _st()))
                if gx is not None:
                    dx = dx + gx
        if weight.requires_grad:
            ws = scratch_zeroed('wgrad', ff, x.device)
            _ck(L.muvo_conv_wgrad(C.byref(d), _f(x), _f(gy), _f(ws), _f(grad_of(weight)), _f(grad_of(bias)) if bias is not None else None,
                                  None, None, 0, _st()))
        return dx, None, None, None, None


class ConvHeadFn(torch.autograd.Function):
    """(y, logits) = (act(conv(x)), head(y)) for a decoder stage whose output feeds a 1x1 head with <= 4 produced channels
    (ConvDecoder: trans_conv1/2/3 + head_4/2/1, common.py:608-632).  Forward: the two convolutions.  Backward: the head's weight /
    bias gradient, then the stage's own backward with the head's data gradient added to dy INSIDE the split pass
    (muvo_conv_prepare_dy_head) — neither the accumulate pass over the feature map (HeadBranchFn) nor, at the last stage, the
    materialised head gradient exists."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, packed, act, slope, head_w, head_b, head_geom, head_packed):
        n0 = x.shape[0]
        in_sz0 = tuple(x.shape[2:]) if geom.nd == 3 else (1,) + tuple(x.shape[2:])
        d0, out_sz0 = geom.plan(n0, in_sz0)[:2]
        co = head_geom.cout
        fkey = ('head_fwd', n0, in_sz0, co, _plan_epoch[0])
        fused = geom.family.get(fkey)
        if fused is None:
            fused = geom.family[fkey] = bool(lib().muvo_conv_forward_head_supported(C.byref(d0), co))
        logits = None
        if fused:
            # the head's forward rides on the stage's epilogue (muvo_conv_forward_head): no pass over y
            logits = torch.empty((n0, co) + (out_sz0 if geom.nd == 3 else out_sz0[1:]), device=x.device, dtype=torch.float32)
            ctx._head_fwd = (head_w.contiguous().view(co, -1), head_b, logits)
        y = ConvFn.forward(ctx, x, weight, bias, geom, packed, act, slope)
        ctx._head_fwd = None
        ctx.save_for_backward(x, y)            # y is needed for the head's weight gradient even when act needs no derivative
        if fused:
            ctx.head = (head_w, head_b, head_geom, tuple(y.shape[2:]) if head_geom.nd == 3 else (1,) + tuple(y.shape[2:]))
            ctx.set_materialize_grads(False)
            return y, logits
        # an output nobody differentiates through (the last stage's feature map has the head as its only consumer) arrives
        # as None in backward instead of a zero tensor of its size (1.36 GB filled, then read by the split pass)
        ctx.set_materialize_grads(False)
        n = y.shape[0]
        hin = tuple(y.shape[2:]) if head_geom.nd == 3 else (1,) + tuple(y.shape[2:])
        hd, hout, hff, _ = head_geom.plan(n, hin)
        L = lib()
        k, pkey = _wkey(head_w), (hin, _plan_epoch[0])
        if _head_alias(head_geom, (n, hin, _plan_epoch[0]), 0, head_w):
            head_packed.fwd, head_packed.fwd_key, head_packed.fwd_plan = head_w.detach().view(-1), k, pkey
        else:
            if head_packed.fwd is None or head_packed.fwd.numel() < hff or _is_alias(head_packed.fwd, head_w):
                head_packed.fwd = torch.empty(hff, device=x.device, dtype=torch.float32)
                head_packed.fwd_key = None
            if head_packed.fwd_key != k or head_packed.fwd_plan != pkey:
                _ck(L.muvo_conv_pack_weights(C.byref(hd), _f(head_w), _f(head_packed.fwd), None, _st()))
                head_packed.fwd_key, head_packed.fwd_plan = k, pkey
                _PACKS.register((id(head_packed), pkey), ('conv', hd, head_w, weakref.ref(head_packed), pkey))
        logits = torch.empty((n, head_geom.cout) + (hout if head_geom.nd == 3 else hout[1:]), device=x.device, dtype=torch.float32)
        _ck(L.muvo_conv_forward(C.byref(hd), _f(y), _f(head_packed.fwd), _f(head_b), _f(logits), ACT_NONE, _fl(0.0), None, _st()))
        ctx.head = (head_w, head_b, head_geom, hin)
        return y, logits

    @staticmethod
    def backward(ctx, gy, gl):
        if gl is None:
            if gy is None:
                return (None,) * 11
            return ConvFn._backward(ctx, gy, None)[:7] + (None,) * 4
        x, y = ctx.saved_tensors
        head_w, head_b, head_geom, hin = ctx.head
        hd, _, hff, _ = head_geom.plan(y.shape[0], hin)
        gl = gl.contiguous()

        def head_wgrad():        # the head's own weight-gradient pass (when it cannot ride on the split pass)
            if head_w.requires_grad:
                ws = scratch_zeroed('wgrad', hff, x.device)
                _ck(lib().muvo_conv_wgrad(C.byref(hd), _f(y), _f(gl), _f(ws), _f(grad_of(head_w)),
                                          _f(grad_of(head_b)) if head_b is not None else None, None, None, 0, _st()))
        return ConvFn._backward(ctx, gy, (gl, head_w, head_b, head_wgrad))[:7] + (None,) * 4


def conv_head_supported(x, geom, head_geom, act_bwd_fused=False):
    """can (stage convolution, 1x1 head) run as ConvHeadFn?  The head must be on the 1x1 head kernels, the stage on the bf16x3
    kernels in both gradient directions (its backward preamble is the split pass that absorbs the head's data gradient)."""
    if not CONV_HEAD or not (x.requires_grad and torch.is_grad_enabled()) or act_bwd_fused:
        return False
    n = x.shape[0]
    in_sz = tuple(x.shape[2:]) if geom.nd == 3 else (1,) + tuple(x.shape[2:])
    d, out_sz, _, _ = geom.plan(n, in_sz)
    fam = geom.family[(n, in_sz, _plan_epoch[0])]
    if not (fam[1] == 1 and fam[2] == 1):
        return False
    head_geom.plan(n, out_sz)
    hf = head_geom.family[(n, out_sz, _plan_epoch[0])]
    return hf[0] == 3 and hf[2] == 3 and bool(lib().muvo_conv_prepare_dy_head_supported(C.byref(d), head_geom.cout))


CONV_HEAD = os.environ.get('MUVO_CONV_HEAD', '1') != '0'


def conv_head(x, weight, bias, geom, packed, act, slope, head_w, head_b, head_geom, head_packed):
    return ConvHeadFn.apply(x, weight, bias, geom, packed, act, slope, head_w, head_b, head_geom, head_packed)


def head_branch(x, weight, bias, geom, packed):
    """(x, head(x)); fused backward when the head runs on the 1x1 head kernels, plain conv otherwise."""
    n = x.shape[0]
    in_sz = tuple(x.shape[2:]) if geom.nd == 3 else (1,) + tuple(x.shape[2:])
    geom.plan(n, in_sz)
    fam = geom.family[(n, in_sz, _plan_epoch[0])]
    if fam[0] == 3 and fam[1] == 3 and fam[2] == 3 and x.requires_grad and torch.is_grad_enabled():
        return HeadBranchFn.apply(x, weight, bias, geom, packed)
    return x, conv(x, weight, bias, geom, packed)


# ================================================================================================ norms
BN_PLANES = os.environ.get('MUVO_BN_PLANES', '1') != '0'


_NAN_SCALAR = {}


def _nan_placeholder(shape, device):
    """a tensor of `shape` without storage of its size (one NaN, expanded): stands for data that exists only as split planes
    (attribute _muvo_planes).  Anything that reads it as numbers gets NaN - a misuse cannot go unnoticed.  The one-element source
    is made once per device (a fresh torch.full per call was 60 fill launches per step); every call returns a new view object, so
    attributes set on one placeholder do not show up on another."""
    device = torch.device(device)
    src = _NAN_SCALAR.get(device)
    if src is None:
        src = _NAN_SCALAR[device] = torch.full((1,), float('nan'), device=device, dtype=torch.float32)
    return src.expand(shape)


class BNActFn(torch.autograd.Function):
    """Train-mode BatchNorm2d + optional residual + ReLU.  res_mode 1: relu(bn(x)+res); 2: relu(bn(x))+res.
    planes: also write the channels-last bf16 hi / lo planes of the result (what a bf16x3 convolution reads;
    muvo_bn_train_fwd_planes) and hand them to the consumers as y._muvo_planes; keep_f32=False: ONLY the planes (y is a
    NaN placeholder) - the caller guarantees that the sole consumer is a convolution that reads planes in forward and weight
    gradient.  dx_planes: the input x is the output of a convolution whose backward reads split planes of its dy
    (x._muvo_dy_planes_ok): backward writes dx as planes only (muvo_bn_train_bwd_planes)."""

    @staticmethod
    def forward(ctx, x, residual, bn, res_mode, relu, training, planes=False, keep_f32=True, dx_planes=False):
        x = x.contiguous()
        n, c = x.shape[:2]
        s = x.numel() // (n * c)
        mean = torch.empty(c, device=x.device, dtype=torch.float32)
        rstd = torch.empty(c, device=x.device, dtype=torch.float32)
        if residual is not None:
            residual = residual.contiguous()
        if not training:
            raise RuntimeError('eval-mode BatchNorm is not part of the training hot path (reference keeps train() mode '
                               'even in validation, trainer.py:405)')
        mask_mode = 0 if not relu else (1 if (residual is not None and res_mode == 1) else 2)
        keep_f32 = keep_f32 or not planes or mask_mode == 1
        y = torch.empty_like(x) if keep_f32 else _nan_placeholder(x.shape, x.device)
        if planes:
            pl = torch.empty((lib().muvo_split_planes_bytes(n, c, _i64(s)) + 3) // 4, device=x.device, dtype=torch.float32)
            _ck(lib().muvo_bn_train_fwd_planes(_f(x), _f(bn.weight), _f(bn.bias), _f(residual), _f(y) if keep_f32 else None, _f(mean),
                                               _f(rstd), _f(bn.running_mean), _f(bn.running_var), n, c, _i64(s), _fl(bn.eps),
                                               _fl(bn.momentum), res_mode if residual is not None else 0, int(relu), _p(pl), _st()))
            _BN_PLANES_OUT.append((pl, _stream_key(x.device), not keep_f32))
        else:
            ws = torch.empty(2 * c, device=x.device, dtype=torch.float64)
            _ck(lib().muvo_bn_train_fwd(_f(x), _f(bn.weight), _f(bn.bias), _f(residual), _f(y), _f(mean), _f(rstd),
                                        _f(bn.running_mean), _f(bn.running_var), _p(ws), n, c, _i64(s), _fl(bn.eps),
                                        _fl(bn.momentum), res_mode if residual is not None else 0, int(relu), _st()))
        _BN_PENDING.append(bn)           # num_batches_tracked += 1, applied in one fused launch (flush_bn_counters)
        ctx.bn, ctx.dims = bn, (n, c, s)
        # ReLU mask in backward: recomputed from x (mode 2: y is neither saved nor re-read) unless the ReLU sits AFTER the
        # residual add (res_mode 1), where only y knows the sign
        ctx.mask_mode = mask_mode
        ctx.has_res, ctx.res_mode = residual is not None, res_mode
        ctx.dx_planes = bool(dx_planes)
        ctx.save_for_backward(x, y if ctx.mask_mode == 1 else None, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean, rstd = ctx.saved_tensors
        bn = ctx.bn
        n, c, s = ctx.dims
        dy = dy.contiguous()
        dres = None
        if ctx.has_res and ctx.needs_input_grad[1]:
            dres = torch.empty_like(x) if ctx.res_mode == 1 else dy
        if ctx.dx_planes:
            # dx leaves as the split planes the producing convolution's data- and weight-gradient kernels read; no fp32 dx
            pl = torch.empty((lib().muvo_split_planes_bytes(n, c, _i64(s)) + 3) // 4, device=x.device, dtype=torch.float32)
            _ck(lib().muvo_bn_train_bwd_planes(_f(x), _f(y), _f(dy), _f(bn.weight), _f(bn.bias), _f(mean), _f(rstd), None,
                                               _f(dres) if (ctx.has_res and ctx.res_mode == 1) else None,
                                               _f(grad_of(bn.weight)), _f(grad_of(bn.bias)), n, c, _i64(s), ctx.mask_mode, _p(pl), _st()))
            dx = _nan_placeholder(x.shape, x.device)
            dx._muvo_planes = (pl, _stream_key(x.device))
            return dx, dres, None, None, None, None, None, None, None
        dx = torch.empty_like(x)
        ws = torch.empty(2 * c, device=x.device, dtype=torch.float64)
        _ck(lib().muvo_bn_train_bwd(_f(x), _f(y), _f(dy), _f(bn.weight), _f(bn.bias), _f(mean), _f(rstd), _f(dx),
                                    _f(dres) if (ctx.has_res and ctx.res_mode == 1) else None,
                                    _f(grad_of(bn.weight)), _f(grad_of(bn.bias)), _p(ws), n, c, _i64(s), ctx.mask_mode,
                                    _st()))
        return dx, dres, None, None, None, None, None, None, None


_BN_PENDING = []


def flush_bn_counters():
    """Apply the pending `num_batches_tracked += 1` of every BatchNorm forward since the last flush with one
    torch._foreach_add_ (76 one-element launches per step otherwise).  Called at the start of every training forward and
    before a BatchNorm's buffers are read (state_dict)."""
    if not _BN_PENDING:
        return
    counts = {}
    for bn in _BN_PENDING:
        counts[id(bn)] = (bn, counts.get(id(bn), (bn, 0))[1] + 1)
    _BN_PENDING.clear()
    by_k = {}
    for bn, k in counts.values():
        by_k.setdefault(k, []).append(bn.num_batches_tracked)
    for k, tensors in by_k.items():
        torch._foreach_add_(tensors, k)


def conv_reads_planes(conv, x_shape, need_wgrad):
    """will the convolution module `conv` read split planes of an input of shape x_shape in forward (and, if a backward follows,
    in its weight gradient)?  -> (forward reads planes, and the fp32 values are needed by nobody inside the convolution)"""
    geom = conv.geom
    n = x_shape[0]
    in_sz = tuple(x_shape[2:]) if geom.nd == 3 else (1,) + tuple(x_shape[2:])
    geom.plan(n, in_sz)
    key = (n, in_sz, _plan_epoch[0])
    fam, wsb = geom.family[key], geom.ws_bytes[key]
    fwd = fam[0] == 1 and wsb[0] > 0
    return fwd, fwd and (not need_wgrad or (fam[2] == 1 and wsb[2] > 0 and _KEEP_WS))


def bn_act(x, bn, residual=None, res_mode=1, relu=True, consumers=(), sole_consumer=False, from_conv=False):
    """consumers: convolution modules that will read the result (planes are written with it when at least one of them runs on the
    bf16x3 kernels); sole_consumer: `consumers` is everything that reads the result, so the fp32 tensor is dropped when they all
    read planes; from_conv: x is the output of a bias-free convolution without activation and this BatchNorm is its only
    consumer (conv -> bn): dx is handed to that convolution's backward as planes when it reads planes."""
    planes, keep_f32 = False, True
    if BN_PLANES and consumers and x.is_cuda:
        grad = torch.is_grad_enabled() and x.requires_grad
        rd = [conv_reads_planes(cv, x.shape, grad and cv.weight.requires_grad) for cv in consumers]
        planes = any(r[0] for r in rd)
        keep_f32 = not (sole_consumer and all(r[1] for r in rd))
    dx_planes = BN_PLANES and from_conv and getattr(x, '_muvo_dy_planes_ok', False)
    y = BNActFn.apply(x, residual, bn, res_mode, relu, bn.training, planes, keep_f32, dx_planes)
    if planes:
        pl = _BN_PLANES_OUT.pop()
        y._muvo_planes = pl[:2]
        if pl[2]:
            y._muvo_planes_only = True
    return y


_BN_PLANES_OUT = []


class AdaINFn(torch.autograd.Function):
    """AdaptiveInstanceNorm3d. x: (N,C,D,H,W) or a broadcast (C,D,H,W) parameter; style: (N, 2C)."""

    @staticmethod
    def forward(ctx, x, style, eps, n_batch, pre_act=ACT_NONE, pre_slope=0.0, moments=None, aff_out=None):
        x = x.contiguous()
        style = style.contiguous()
        bcast = x.dim() == 4
        ctx.pre_act, ctx.pre_slope = pre_act, pre_slope
        c = x.shape[0] if bcast else x.shape[1]
        s = x.numel() // c if bcast else x.numel() // (x.shape[0] * c)
        n = n_batch
        mean = torch.empty(n * c, device=x.device, dtype=torch.float32)
        rstd = torch.empty(n * c, device=x.device, dtype=torch.float32)
        if aff_out is not None:
            # lazy form: statistics -> (mean, rstd, scale / shift table); the consumer applies the map while staging x, the
            # normalised tensor is never written.  The returned tensor is a shape-only placeholder (no storage of its size).
            _ck(lib().muvo_adain_affine(_f(style), _p(moments), _f(mean), _f(rstd), _f(aff_out), n, c, _i64(s), _fl(eps), _st()))
            ctx.dims = (n, c, s, bcast)
            ctx.save_for_backward(x, style, mean, rstd)
            return torch.empty(1, device=x.device, dtype=torch.float32).expand((n, c) + tuple(x.shape[-3:]))
        y = torch.empty((n, c) + tuple(x.shape[-3:]), device=x.device, dtype=torch.float32)
        if moments is not None and not bcast:   # statistics already accumulated by the producing convolution's epilogue
            _ck(lib().muvo_adain_fwd_moments(_f(x), _f(style), _f(y), _f(mean), _f(rstd), _p(moments), n, c, _i64(s), _fl(eps), _st()))
        else:
            ws = torch.empty(2 * n * c, device=x.device, dtype=torch.float64)
            _ck(lib().muvo_adain_fwd(_f(x), _f(style), _f(y), _f(mean), _f(rstd), _p(ws), n, c, _i64(s),
                                     _i64(0 if bcast else c * s), _fl(eps), _st()))
        ctx.dims = (n, c, s, bcast)
        ctx.save_for_backward(x, style, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, style, mean, rstd = ctx.saved_tensors
        n, c, s, bcast = ctx.dims
        dy = dy.contiguous()
        dxf = torch.empty_like(dy)
        dstyle = torch.empty_like(style)
        ws = torch.empty(2 * n * c, device=x.device, dtype=torch.float64)
        _ck(lib().muvo_adain_bwd(_f(x), _f(style), _f(dy), _f(mean), _f(rstd), _f(dxf), _f(dstyle), _p(ws), n, c,
                                 _i64(s), _i64(0 if bcast else c * s), ctx.pre_act, _fl(ctx.pre_slope), _st()))
        if bcast:
            dx = torch.empty_like(x)
            _ck(lib().muvo_batchsum(_f(dxf), _f(dx), n, _i64(c * s), 0, _st()))
        else:
            dx = dxf
        return dx, dstyle, None, None, None, None, None, None


def adain_lazy(x, style, eps, pre_act, pre_slope, moments):
    """AdaIN of x (N, C, D, H, W) whose consumer applies it while staging: returns (placeholder, x, aff) for
    conv(..., lazy=(x, aff)).  The placeholder carries the autograd edge and the shape, nothing else may read it."""
    n, c = x.shape[:2]
    aff = torch.empty(n, c, 2, device=x.device, dtype=torch.float32)
    y = AdaINFn.apply(x, style, eps, n, pre_act, pre_slope, moments, aff)
    return y, x, aff


def adain(x, style, eps, n_batch, pre_act=ACT_NONE, pre_slope=0.0, moments=None):
    """pre_act: x is the output of that activation and the producer's backward does NOT apply its derivative (the
    AdaIN backward kernel chains it); pair with conv(..., act_bwd_fused=True)."""
    return AdaINFn.apply(x, style, eps, n_batch, pre_act, pre_slope, moments)


class AdaINHeadFn(torch.autograd.Function):
    """AdaIN of the last voxel-decoder convolution + the 1x1x1 class head in one pass each way (csrc/norm.hip:
    adain_head_*): x (N,C,D,H,W) is the convolution output (LeakyReLU applied, its derivative chained here), `moments` its
    per-(n,c) sums from the convolution epilogue.  The normalised tensor never exists in HBM."""

    @staticmethod
    def forward(ctx, x, style, head_w, head_b, eps, moments, pre_act, pre_slope):
        x, style = x.contiguous(), style.contiguous()
        n, c = x.shape[:2]
        s = x.numel() // (n * c)
        co = head_w.shape[0]
        logits = torch.empty((n, co) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
        mean = torch.empty(n * c, device=x.device, dtype=torch.float32)
        rstd = torch.empty(n * c, device=x.device, dtype=torch.float32)
        _ck(lib().muvo_adain_head_fwd(_f(x), _f(style), _f(mean), _f(rstd), _p(moments), _f(head_w.contiguous().view(co, c)),
                                      _f(head_b), _f(logits), n, c, co, _i64(s), _fl(eps), _st()))
        ctx.dims, ctx.pre = (n, c, co, s), (pre_act, pre_slope)
        ctx.head_w, ctx.head_b = head_w, head_b
        ctx.save_for_backward(x, style, mean, rstd)
        return logits

    @staticmethod
    def backward(ctx, dl):
        x, style, mean, rstd = ctx.saved_tensors
        n, c, co, s = ctx.dims
        head_w, head_b = ctx.head_w, ctx.head_b
        dl = dl.contiguous()
        dx = torch.empty_like(x)
        dstyle = torch.empty_like(style)
        ws = torch.empty(2 * n * c, device=x.device, dtype=torch.float64)
        _ck(lib().muvo_adain_head_bwd(_f(x), _f(style), _f(mean), _f(rstd), _f(head_w.contiguous().view(co, c)), _f(dl), _f(dx),
                                      _f(dstyle), _f(grad_of(head_w)), _f(grad_of(head_b)) if head_b is not None else None, _p(ws),
                                      n, c, co, _i64(s), ctx.pre[0], _fl(ctx.pre[1]), _st()))
        return dx, dstyle, None, None, None, None, None, None


ADAIN_HEAD = os.environ.get('MUVO_ADAIN_HEAD', '1') != '0'


def adain_head_supported(x, head_w, moments):
    if not ADAIN_HEAD or moments is None or x.dim() != 5:
        return False
    n, c = x.shape[:2]
    return bool(lib().muvo_adain_head_supported(c, head_w.shape[0], _i64(x.numel() // (n * c)))) and n <= 65535


def adain_head(x, style, head_w, head_b, eps, moments, pre_act=ACT_NONE, pre_slope=0.0):
    return AdaINHeadFn.apply(x, style, head_w, head_b, eps, moments, pre_act, pre_slope)


class AddDropoutLNFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(a)) over the last dim."""

    @staticmethod
    def forward(ctx, x, a, ln, p, seed):
        x = x.contiguous()
        a = a.contiguous()
        e = x.shape[-1]
        rows = x.numel() // e
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
        _ck(lib().muvo_add_dropout_layernorm_fwd(_f(x), _f(a), _f(ln.weight), _f(ln.bias), _f(y), _f(z), _f(mean),
                                                 _f(rstd), rows, e, _fl(ln.eps), _fl(p), C.c_uint64(seed), _st()))
        ctx.ln, ctx.p, ctx.seed, ctx.dims = ln, p, seed, (rows, e)
        ctx.save_for_backward(z, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd = ctx.saved_tensors
        ln = ctx.ln
        rows, e = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        da = torch.empty_like(dy)
        _ck(lib().muvo_add_dropout_layernorm_bwd(_f(dy), _f(z), _f(mean), _f(rstd), _f(ln.weight), _f(dx), _f(da),
                                                 _f(grad_of(ln.weight)), _f(grad_of(ln.bias)), rows, e, _fl(ctx.p),
                                                 C.c_uint64(ctx.seed), _st()))
        return dx, da, None, None, None


def add_dropout_layernorm(x, a, ln, p, seed):
    return AddDropoutLNFn.apply(x, a, ln, p, seed)


# ================================================================================================ misc
class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        x = x.contiguous()
        y = torch.empty_like(x)
        _ck(lib().muvo_act_fwd(_f(x), _f(y), _i64(x.numel()), act, _fl(slope), _st()))
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _ck(lib().muvo_act_bwd(_f(y), _f(dy), _f(dx), _i64(dy.numel()), ctx.act, _fl(ctx.slope), _st()))
        return dx, None, None


def activation(x, act, slope=0.0):
    return ActFn.apply(x, act, slope)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.contiguous()
        y = torch.empty_like(x)
        _ck(lib().muvo_dropout(_f(x), _f(y), _i64(x.numel()), _fl(p), C.c_uint64(seed), _st()))
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _ck(lib().muvo_dropout(_f(dy), _f(dx), _i64(dy.numel()), _fl(ctx.p), C.c_uint64(ctx.seed), _st()))
        return dx, None, None


def dropout(x, p, seed):
    if p <= 0.0:
        return x
    return DropoutFn.apply(x, p, seed)


class CatLastFn(torch.autograd.Function):
    """torch.cat(tensors, dim=-1) for 2-D row-major tensors, via strided 2-D copies."""

    @staticmethod
    def forward(ctx, *xs):
        rows = xs[0].shape[0]
        widths = [x.shape[1] for x in xs]
        tot = sum(widths)
        y = torch.empty(rows, tot, device=xs[0].device, dtype=torch.float32)
        off = 0
        for x, w in zip(xs, widths):
            x = x.contiguous()
            _ck(lib().muvo_copy2d(_f(x), C.c_void_p(y.data_ptr() + 4 * off), _i64(rows), _i64(w), _i64(w), _i64(tot), 0,
                                  _st()))
            off += w
        ctx.widths, ctx.rows = widths, rows
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        tot = sum(ctx.widths)
        outs, off = [], 0
        for i, w in enumerate(ctx.widths):
            if ctx.needs_input_grad[i]:
                g = torch.empty(ctx.rows, w, device=dy.device, dtype=torch.float32)
                _ck(lib().muvo_copy2d(C.c_void_p(dy.data_ptr() + 4 * off), _f(g), _i64(ctx.rows), _i64(w), _i64(tot),
                                      _i64(w), 0, _st()))
                outs.append(g)
            else:
                outs.append(None)
            off += w
        return tuple(outs)


def cat_last(xs):
    return CatLastFn.apply(*xs)


class SliceLastFn(torch.autograd.Function):
    """x[:, a:b] as a fresh contiguous tensor (2-D)."""

    @staticmethod
    def forward(ctx, x, a, b):
        x = x.contiguous()
        rows, tot = x.shape
        y = torch.empty(rows, b - a, device=x.device, dtype=torch.float32)
        _ck(lib().muvo_copy2d(C.c_void_p(x.data_ptr() + 4 * a), _f(y), _i64(rows), _i64(b - a), _i64(tot), _i64(b - a), 0,
                              _st()))
        ctx.a, ctx.b, ctx.tot, ctx.rows = a, b, tot, rows
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.zeros(ctx.rows, ctx.tot, device=dy.device, dtype=torch.float32)
        _ck(lib().muvo_copy2d(_f(dy), C.c_void_p(dx.data_ptr() + 4 * ctx.a), _i64(ctx.rows), _i64(ctx.b - ctx.a),
                              _i64(ctx.b - ctx.a), _i64(ctx.tot), 0, _st()))
        return dx, None, None


def slice_last(x, a, b):
    return SliceLastFn.apply(x, a, b)


class TokensFn(torch.autograd.Function):
    """(N,C,h,w) feature maps of the two sensors -> (L_img+L_lidar, N, C) tokens with positional + type embedding."""

    @staticmethod
    def forward(ctx, xi, xl, pos_i, pos_l, type_emb):
        xi, xl = xi.contiguous(), xl.contiguous()
        n, c = xi.shape[:2]
        li, ll = xi.shape[2] * xi.shape[3], xl.shape[2] * xl.shape[3]
        tok = torch.empty(li + ll, n, c, device=xi.device, dtype=torch.float32)
        L = lib()
        te = type_emb.contiguous()  # (1,1,C,2): element [c][k] at c*2+k
        _ck(L.muvo_nchw_to_tokens(_f(xi), _f(pos_i), _f(te), 2, _f(tok), n, c, li, 0, _st()))
        _ck(L.muvo_nchw_to_tokens(_f(xl), _f(pos_l), C.c_void_p(te.data_ptr() + 4), 2, _f(tok), n, c, ll, li, _st()))
        ctx.shapes = (xi.shape, xl.shape, li, ll, n, c)
        ctx.type_emb = type_emb
        return tok

    @staticmethod
    def backward(ctx, dtok):
        dtok = dtok.contiguous()
        si, sl, li, ll, n, c = ctx.shapes
        L = lib()
        dxi = torch.empty(si, device=dtok.device, dtype=torch.float32)
        dxl = torch.empty(sl, device=dtok.device, dtype=torch.float32)
        _ck(L.muvo_tokens_to_nchw(_f(dtok), _f(dxi), n, c, li, 0, _st()))
        _ck(L.muvo_tokens_to_nchw(_f(dtok), _f(dxl), n, c, ll, li, _st()))
        te = ctx.type_emb
        if te.requires_grad:
            tmp = torch.zeros(2, c, device=dtok.device, dtype=torch.float32)
            _ck(L.muvo_colsum_acc(_f(dtok), _f(tmp[0]), _i64(li * n), _i64(c), _i64(c), _st()))
            _ck(L.muvo_colsum_acc(C.c_void_p(dtok.data_ptr() + 4 * li * n * c), _f(tmp[1]), _i64(ll * n), _i64(c),
                                  _i64(c), _st()))
            # grad layout (1,1,C,2): interleave the two columns with strided copies
            g = grad_of(te)
            for k in range(2):
                _ck(L.muvo_copy2d(_f(tmp[k]), C.c_void_p(g.data_ptr() + 4 * k), _i64(c), _i64(1), _i64(1), _i64(2), 1,
                                  _st()))
        return dxi, dxl, None, None, None


def make_tokens(xi, xl, pos_i, pos_l, type_emb):
    return TokensFn.apply(xi, xl, pos_i, pos_l, type_emb)


class UntokenFn(torch.autograd.Function):
    """tokens[l0:l0+h*w] (L,N,C) -> (N,C,h,w)."""

    @staticmethod
    def forward(ctx, tok, l0, h, w):
        tok = tok.contiguous()
        ltot, n, c = tok.shape
        x = torch.empty(n, c, h, w, device=tok.device, dtype=torch.float32)
        _ck(lib().muvo_tokens_to_nchw(_f(tok), _f(x), n, c, h * w, l0, _st()))
        ctx.dims = (ltot, n, c, l0, h * w)
        return x

    @staticmethod
    def backward(ctx, dx):
        dx = dx.contiguous()
        ltot, n, c, l0, l = ctx.dims
        dtok = torch.zeros(ltot, n, c, device=dx.device, dtype=torch.float32)
        _ck(lib().muvo_nchw_to_tokens(_f(dx), None, None, 0, _f(dtok), n, c, l, l0, _st()))
        return dtok, None, None, None


def untoken(tok, l0, h, w):
    return UntokenFn.apply(tok, l0, h, w)


class MaxPool2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        x = x.contiguous()
        n, c, h, w = x.shape
        oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        y = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float32)
        idx = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.uint8)
        _ck(lib().muvo_maxpool2d_fwd(_f(x), _f(y), _p(idx), _i64(n * c), h, w, oh, ow, k, s, p, _st()))
        ctx.dims = (n, c, h, w, oh, ow, k, s, p)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w, oh, ow, k, s, p = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty(n, c, h, w, device=dy.device, dtype=torch.float32)
        _ck(lib().muvo_maxpool2d_bwd(_f(dy), _p(idx), _f(dx), _i64(n * c), h, w, oh, ow, k, s, p, _st()))
        return dx, None, None, None


def max_pool2d(x, k, s=None, p=0):
    return MaxPool2dFn.apply(x, k, s or k, p)


class GlobalAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, c = x.shape[:2]
        s = x.numel() // (n * c)
        y = torch.empty(n, c, device=x.device, dtype=torch.float32)
        _ck(lib().muvo_avgpool_fwd(_f(x), _f(y), _i64(n * c), _i64(s), _st()))
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        shape = ctx.shape
        dx = torch.empty(shape, device=dy.device, dtype=torch.float32)
        g = shape[0] * shape[1]
        _ck(lib().muvo_avgpool_bwd(_f(dy), _f(dx), _i64(g), _i64(dx.numel() // g), _st()))
        return dx


def global_avg_pool(x):
    return GlobalAvgPoolFn.apply(x)


class Upsample3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, c, d, h, w = x.shape
        y = torch.empty(n, c, 2 * d, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
        _ck(lib().muvo_upsample3d_x2_fwd(_f(x), _f(y), _i64(n * c), d, h, w, _st()))
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        n, c, d, h, w = ctx.shape
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        _ck(lib().muvo_upsample3d_x2_bwd(_f(dy), _f(dx), _i64(n * c), d, h, w, _st()))
        return dx


def upsample3d_x2(x):
    return Upsample3dFn.apply(x)


# ================================================================================================ attention
class AttentionFn(torch.autograd.Function):
    """Multi-head self-attention core on a packed (L, N, 3E) qkv tensor -> (L, N, E).

    QK^T and PV are strided batched MFMA GEMMs straight on the packed layout (no head split copies);
    softmax + attention-probability dropout is one wave-per-row kernel."""

    @staticmethod
    def forward(ctx, qkv, nheads, p, seed):
        qkv = qkv.contiguous()
        l, n, e3 = qkv.shape
        e = e3 // 3
        dh = e // nheads
        scale = 1.0 / (dh ** 0.5)
        dev = qkv.device
        S = torch.empty(n, nheads, l, l, device=dev, dtype=torch.float32)
        # S[n][h][i][j] = scale * sum_d q[i][n][h*dh+d] * k[j][n][e + h*dh + d]
        gemm(qkv, qkv, S, l, l, dh, n * e3, 1, 1, n * e3, l, alpha=scale, B1=n, B2=nheads,
             a_b=(e3, dh), b_b=(e3, dh), c_b=(nheads * l * l, l * l), b_off=e)
        P = torch.empty_like(S)
        Pd = torch.empty_like(S) if p > 0 else None
        _ck(lib().muvo_softmax_dropout_fwd(_f(S), _f(P), _f(Pd), _i64(n * nheads * l), l, _fl(p), C.c_uint64(seed),
                                           _st()))
        del S
        o = torch.empty(l, n, e, device=dev, dtype=torch.float32)
        pa = Pd if Pd is not None else P
        # o[i][n][h*dh+d] = sum_j P[n][h][i][j] * v[j][n][2e + h*dh + d]
        gemm(pa, qkv, o, l, dh, l, l, 1, n * e3, 1, n * e, B1=n, B2=nheads, a_b=(nheads * l * l, l * l),
             b_b=(e3, dh), c_b=(e, dh), b_off=2 * e)
        ctx.dims = (l, n, e, nheads, dh, scale, p, seed)
        ctx.save_for_backward(qkv, P)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, P = ctx.saved_tensors
        l, n, e, nheads, dh, scale, p, seed = ctx.dims
        e3 = 3 * e
        do = do.contiguous()
        dev = do.device
        dqkv = torch.empty_like(qkv)
        if p > 0:
            Pd = torch.empty_like(P)
            _ck(lib().muvo_dropout(_f(P), _f(Pd), _i64(P.numel()), _fl(p), C.c_uint64(seed), _st()))
        else:
            Pd = P
        bP = (nheads * l * l, l * l)
        # dV[j][n][h*dh+d] = sum_i Pd[i][j] * do[i][n][h*dh+d]   (A(m=j,k=i)=Pd[i*l+j])
        gemm(Pd, do, dqkv, l, dh, l, 1, l, n * e, 1, n * e3, B1=n, B2=nheads, a_b=bP, b_b=(e, dh), c_b=(e3, dh),
             c_off=2 * e)
        # dPd[i][j] = sum_d do[i][n][h*dh+d] * v[j][n][2e+h*dh+d]
        dPd = torch.empty_like(P)
        gemm(do, qkv, dPd, l, l, dh, n * e, 1, 1, n * e3, l, B1=n, B2=nheads, a_b=(e, dh), b_b=(e3, dh), c_b=bP,
             b_off=2 * e)
        dS = dPd  # in place
        _ck(lib().muvo_softmax_dropout_bwd(_f(P), _f(dPd), _f(dS), _i64(n * nheads * l), l, _fl(p), C.c_uint64(seed),
                                           _st()))
        # dQ[i][.] = scale * sum_j dS[i][j] * k[j][.]
        gemm(dS, qkv, dqkv, l, dh, l, l, 1, n * e3, 1, n * e3, alpha=scale, B1=n, B2=nheads, a_b=bP, b_b=(e3, dh),
             c_b=(e3, dh), b_off=e)
        # dK[j][.] = scale * sum_i dS[i][j] * q[i][.]
        gemm(dS, qkv, dqkv, l, dh, l, 1, l, n * e3, 1, n * e3, alpha=scale, B1=n, B2=nheads, a_b=bP, b_b=(e3, dh),
             c_b=(e3, dh), c_off=e)
        return dqkv, None, None, None


class FlashAttentionFn(torch.autograd.Function):
    """The same attention core as ONE kernel forward and two backward (csrc/attention.hip): K / V of a head staged in LDS,
    softmax over register-resident score tiles with wave shuffles, no L x L tensor in HBM; saves only the row log-sum-exp."""

    @staticmethod
    def forward(ctx, qkv, nheads, p, seed):
        qkv = qkv.contiguous()
        l, n, e3 = qkv.shape
        e = e3 // 3
        o = torch.empty(l, n, e, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(n * nheads, l, device=qkv.device, dtype=torch.float32)
        _ck(lib().muvo_attention_fwd(_f(qkv), _f(o), _f(lse), l, n, nheads, e // nheads, _fl(p), C.c_uint64(seed), _st()))
        ctx.dims = (l, n, nheads, e // nheads, p, seed)
        ctx.save_for_backward(qkv, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        l, n, nheads, dh, p, seed = ctx.dims
        dqkv = torch.empty_like(qkv)
        _ck(lib().muvo_attention_bwd(_f(qkv), _f(o), _f(do.contiguous()), _f(lse), _f(dqkv), l, n, nheads, dh, _fl(p),
                                     C.c_uint64(seed), _st()))
        return dqkv, None, None, None


FLASH_ATTENTION = os.environ.get('MUVO_FLASH_ATTN', '1') != '0'


def attention(qkv, nheads, p, seed):
    l, _, e3 = qkv.shape
    if FLASH_ATTENTION and lib().muvo_attention_supported(l, e3 // 3 // nheads):
        return FlashAttentionFn.apply(qkv, nheads, p, seed)
    return AttentionFn.apply(qkv, nheads, p, seed)


# ================================================================================================ RSSM
class GRUPointwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, gh, h):
        gi, gh, h = gi.contiguous(), gh.contiguous(), h.contiguous()
        b, hd = h.shape
        hn = torch.empty_like(h)
        _ck(lib().muvo_gru_fwd(_f(gi), _f(gh), _f(h), _f(hn), b, hd, _st()))
        ctx.save_for_backward(gi, gh, h)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        gi, gh, h = ctx.saved_tensors
        b, hd = h.shape
        dhn = dhn.contiguous()
        dgi, dgh, dh = torch.empty_like(gi), torch.empty_like(gh), torch.empty_like(h)
        _ck(lib().muvo_gru_bwd(_f(gi), _f(gh), _f(h), _f(dhn), _f(dgi), _f(dgh), _f(dh), b, hd, _st()))
        return dgi, dgh, dh


def gru_pointwise(gi, gh, h):
    return GRUPointwiseFn.apply(gi, gh, h)


class RSSMSampleFn(torch.autograd.Function):
    """(mu|log_sigma) -> mu, sigma = 2*sigmoid(ls/2)+0.1, sample = mu + sigma*eps."""

    @staticmethod
    def forward(ctx, mls, eps, min_std):
        mls = mls.contiguous()
        b, s2 = mls.shape
        s = s2 // 2
        mu, sigma, sample = (torch.empty(b, s, device=mls.device, dtype=torch.float32) for _ in range(3))
        if eps is not None:
            assert eps.stride(-1) == 1 and eps.shape == (b, s)
            eld = eps.stride(0)
            pe = C.c_void_p(eps.data_ptr())
        else:
            eld, pe = 0, None
        _ck(lib().muvo_rssm_sample_fwd(_f(mls), pe, _i64(eld), _f(mu), _f(sigma), _f(sample), b, s, _fl(min_std), _st()))
        ctx.eps = eps
        ctx.save_for_backward(mls)
        return mu, sigma, sample

    @staticmethod
    def backward(ctx, dmu, dsigma, dsample):
        (mls,) = ctx.saved_tensors
        b, s2 = mls.shape
        s = s2 // 2
        eps = ctx.eps
        if eps is not None:
            eld, pe = eps.stride(0), C.c_void_p(eps.data_ptr())
        else:
            eld, pe = 0, None

        def cg(t):
            return None if t is None else t.contiguous()
        dmu, dsigma, dsample = cg(dmu), cg(dsigma), cg(dsample)
        dmls = torch.empty_like(mls)
        _ck(lib().muvo_rssm_sample_bwd(_f(mls), pe, _i64(eld), _f(dmu), _f(dsigma), _f(dsample), _f(dmls), b, s, _st()))
        return dmls, None, None


def rssm_sample(mls, eps, min_std=0.1):
    return RSSMSampleFn.apply(mls, eps, min_std)


def _ptr_table(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if t is not None:
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
            arr[i] = t.data_ptr()
    return arr


def rssm_weights(rssm):
    """The 18 parameter tensors in the order of include/muvo_hip.h: muvo_rssm_forward."""
    return [rssm.pre_gru_net[0].weight, rssm.pre_gru_net[0].bias, rssm.recurrent_model.weight_ih, rssm.recurrent_model.weight_hh,
            rssm.recurrent_model.bias_ih, rssm.recurrent_model.bias_hh, rssm.prior_action_module[0].weight,
            rssm.prior_action_module[0].bias, rssm.posterior_action_module[0].weight, rssm.posterior_action_module[0].bias,
            rssm.prior.module[0].weight, rssm.prior.module[0].bias, rssm.prior.module[2].weight, rssm.prior.module[2].bias,
            rssm.posterior.module[0].weight, rssm.posterior.module[0].bias, rssm.posterior.module[2].weight,
            rssm.posterior.module[2].bias]


class RSSMFusedFn(torch.autograd.Function):
    """RSSM.forward (transition.py:76-127) for a whole sequence: one persistent kernel forward, one backward (csrc/rssm.hip),
    then one skinny GEMM per weight for dW = dY^T X over the b*T rows."""

    @staticmethod
    def forward(ctx, emb, act, noise, rssm, mask):
        emb, act, noise = emb.contiguous(), act.contiguous(), noise.contiguous()
        B, T, E = emb.shape
        AD, H, S, A = act.shape[-1], rssm.hidden_state_dim, rssm.state_dim, rssm.action_latent_dim
        dev = emb.device
        new = lambda n: torch.empty(B, T, n, device=dev, dtype=torch.float32)
        out = [new(H)] + [new(S) for _ in range(6)]
        keep = [new(n) for n in (H, S, AD, H, 3 * H, 3 * H, H + A, H + E + A, H + A, H + E + A, 2 * S, 2 * S)]
        bar = rssm_barrier_words(dev)
        rssm_check(dev, post=False)
        if get_deterministic():
            # deterministic mode runs single-workgroup weight-gradient kernels on the side streams that can hold a compute unit for
            # a long time: the persistent grid must not wait for them inside its bounded barrier spin
            join_side_streams(dev)
        w = rssm_weights(rssm)
        _ck(lib().muvo_rssm_forward(B, T, H, S, E, A, AD, _ptr_table(w), _f(emb), _f(act), _f(noise), C.c_uint64(mask),
                                    _ptr_table(out), _ptr_table(keep), _p(bar), _fl(rssm.prior.min_std), _st()))
        if not any(ctx.needs_input_grad):
            rssm_check(dev)        # no backward will follow (validation / imagination): post the error-word copy here
        ctx.rssm, ctx.mask, ctx.dims = rssm, mask, (B, T, H, S, E, A, AD)
        ctx.save_for_backward(noise, *keep)
        return tuple(out)

    @staticmethod
    def backward(ctx, *gout):
        noise, hprev, zprev, aprev, u, gi, gh, xp, xq, y1p, y1q, mls_p, mls_q = ctx.saved_tensors
        B, T, H, S, E, A, AD = ctx.dims
        rssm = ctx.rssm
        dev = noise.device
        L = lib()
        gout = [None if g is None else g.contiguous() for g in gout]
        new = lambda n: torch.empty(B, T, n, device=dev, dtype=torch.float32)
        d_emb = new(E)
        dmls_p, dmls_q, dy1p, dy1q, dgi, dgh, du, dla_p, dla_q = (new(n) for n in (2 * S, 2 * S, H + A, H + E + A, 3 * H, 3 * H, H, A, A))
        wt = scratch('rssm_wt', L.muvo_rssm_transposed_floats(H, S, E, A), dev)
        sc = scratch('rssm_scratch', L.muvo_rssm_scratch_floats(B, H, S, E, A), dev)
        bar = rssm_barrier_words(dev)
        if get_deterministic():
            join_side_streams(dev)
        w = rssm_weights(rssm)
        _ck(L.muvo_rssm_backward(B, T, H, S, E, A, AD, _ptr_table(w), _f(wt), _f(noise), C.c_uint64(ctx.mask),
                                 _ptr_table([hprev, gi, gh, mls_p, mls_q]), _ptr_table(gout),
                                 _ptr_table([d_emb, dmls_p, dmls_q, dy1p, dy1q, dgi, dgh, du, dla_p, dla_q]), _f(sc), _p(bar), _st()))
        rssm_check(dev)
        rows = B * T
        for dz, x, wi in ((du, zprev, 0), (dgi, u, 2), (dgh, hprev, 3), (dla_p, aprev, 6), (dla_q, aprev, 8), (dy1p, xp, 10),
                          (dmls_p, y1p, 12), (dy1q, xq, 14), (dmls_q, y1q, 16)):
            weight = w[wi]
            out_f, in_f = weight.shape
            if weight.requires_grad:
                gemm(dz, x, grad_of(weight), out_f, in_f, rows, 1, out_f, in_f, 1, in_f, mode=1)
        for dz, bi in ((du, 1), (dgi, 4), (dgh, 5), (dla_p, 7), (dla_q, 9), (dy1p, 11), (dmls_p, 13), (dy1q, 15), (dmls_q, 17)):
            if w[bi].requires_grad:
                n = w[bi].numel()
                _ck(L.muvo_colsum_acc(_f(dz), _f(grad_of(w[bi])), _i64(rows), _i64(n), _i64(n), _st()))
        return d_emb, None, None, None, None


FUSED_RSSM = os.environ.get('MUVO_FUSED_RSSM', '1') != '0'
_rssm_watch = {}


def rssm_barrier_words(dev):
    """The four 32-bit words the persistent RSSM kernels synchronise through: [0] the grid-barrier counter (reset by every
    launch), [1] the STICKY error word a workgroup raises when its barrier spin times out (the grid was not co-resident)."""
    t = _scratch.get(('rssm_bar', dev, torch.int32))
    if t is None:
        t = _scratch[('rssm_bar', dev, torch.int32)] = torch.zeros(4, device=dev, dtype=torch.int32)
    return t


def rssm_check(dev, post=True):
    """Raise if an earlier fused RSSM launch on `dev` gave up in its grid barrier.  No synchronisation: after a launch
    (`post`) the error word is copied to pinned host memory behind the kernel, an event marks the copy, and a later call looks
    at the host copy once that event has completed - the error surfaces at the latest one step after the failed launch."""
    w = _rssm_watch.get(dev)
    if w is None:
        w = _rssm_watch[dev] = dict(host=torch.zeros(1, dtype=torch.int32).pin_memory(), ev=None)
    if w['ev'] is not None and w['ev'].query():
        if int(w['host'][0]) != 0:
            # the word is sticky on the device (every later fused launch would leave at its first barrier): clear it and the
            # host copy, so that a caller who catches this and switches to MUVO_FUSED_RSSM=0 / a smaller grid can go on
            w['host'].zero_()
            w['ev'] = None
            rssm_barrier_words(dev)[1:2].zero_()
            raise RuntimeError('muvo_rssm: a persistent RSSM kernel timed out in its grid barrier (its workgroups were not '
                               'co-resident: another persistent kernel or a CU mask is holding compute units); set '
                               'MUVO_FUSED_RSSM=0 or lower MUVO_RSSM_GRID')
        w['ev'] = None
    if post and w['ev'] is None:
        w['host'].copy_(rssm_barrier_words(dev)[1:2], non_blocking=True)
        w['ev'] = torch.cuda.Event()
        w['ev'].record(torch.cuda.current_stream(dev))


def rssm_fused_supported(B, T, H, S, E, A, AD):
    return FUSED_RSSM and bool(lib().muvo_rssm_supported(B, T, H, S, E, A, AD))


def rssm_fused(emb, act, noise, rssm, use_prior):
    mask = sum(1 << t for t, f in enumerate(use_prior) if f)
    return RSSMFusedFn.apply(emb, act, noise, rssm, mask)


# ================================================================================================ preprocess (no grad)
def preprocess_image(img_u8, crop, mean, std):
    """(B,S,3,H,W) u8 -> (label (B,S,3,h,w) in [0,1], normalised image)."""
    img_u8 = img_u8.contiguous()
    b, s, c, h, w = img_u8.shape
    left, top, right, bottom = crop
    ch, cw = bottom - top, right - left
    label = torch.empty(b, s, c, ch, cw, device=img_u8.device, dtype=torch.float32)
    norm = torch.empty_like(label)
    m = (C.c_float * 3)(*mean)
    sd = (C.c_float * 3)(*std)
    _ck(lib().muvo_preprocess_image(_p(img_u8), _f(label), _f(norm), _i64(b * s * c), c, h, w, top, left, ch, cw, m, sd,
                                    _st()))
    return label, norm


def resize_bilinear_aa(x, oh, ow, mean=None, std=None):
    """(..., C, H, W) float -> antialiased linear resize to (oh, ow) (torchvision resize(antialias=True)); with mean / std also
    the ImageNet-normalised copy (returns (y, ynorm))."""
    x = x.contiguous()
    c, h, w = x.shape[-3:]
    y = torch.empty(*x.shape[:-2], oh, ow, device=x.device, dtype=torch.float32)
    yn = torch.empty_like(y) if mean is not None else None
    m = (C.c_float * 3)(*mean) if mean is not None else None
    sd = (C.c_float * 3)(*std) if mean is not None else None
    _ck(lib().muvo_resize_bilinear_aa(_f(x), _f(y), _f(yn), m, sd, _i64(x.numel() // (h * w)), c, h, w, oh, ow, _st()))
    return (y, yn) if mean is not None else y


def preprocess_route(route_u8, size, mean, std):
    route_u8 = route_u8.contiguous()
    b, s, c, h, w = route_u8.shape
    out = torch.empty(b, s, c, size, size, device=route_u8.device, dtype=torch.float32)
    m = (C.c_float * 3)(*mean)
    sd = (C.c_float * 3)(*std)
    _ck(lib().muvo_preprocess_route(_p(route_u8), _f(out), _i64(b * s * c), c, h, w, size, size, m, sd, _st()))
    return out


PIXAUG_STRIDE, ROUTEAUG_STRIDE = 16, 8


def pixel_augment(label, norm, params, mean, std):
    """PixelAugmentation (preprocess.py:295-333) on the cropped [0,1] image `label` (b,s,3,h,w), IN PLACE (it is rgb_label_1),
    re-normalising the augmented frames into `norm`.  params: (b*s, PIXAUG_STRIDE) float32 device table (muvo_amd/augment.py)."""
    b, s, c, h, w = label.shape
    assert c == 3 and label.is_contiguous() and norm.is_contiguous() and params.shape == (b * s, PIXAUG_STRIDE)
    tmp = scratch('pixaug_tmp', label.numel(), label.device)
    gsum = scratch('pixaug_gray', b * s, label.device, torch.float64)
    m = (C.c_float * 3)(*mean)
    sd = (C.c_float * 3)(*std)
    _ck(lib().muvo_pixel_augment(_f(label), _f(norm), _f(tmp), _f(params), _p(gsum), _i64(b * s), h, w, m, sd, _st()))


def preprocess_route_aug(route_u8, size, mean, std, params=None):
    """preprocess_route + RouteAugmentation (preprocess.py:336-367); params: (b, ROUTEAUG_STRIDE) float32 device table or None."""
    route_u8 = route_u8.contiguous()
    b, s, c, h, w = route_u8.shape
    out = torch.empty(b, s, c, size, size, device=route_u8.device, dtype=torch.float32)
    m = (C.c_float * 3)(*mean)
    sd = (C.c_float * 3)(*std)
    assert params is None or params.shape == (b, ROUTEAUG_STRIDE)
    _ck(lib().muvo_preprocess_route_aug(_p(route_u8), _f(out), _f(params), b, s, c, h, w, size, size, m, sd, _st()))
    return out


def divide_scalar(x, divisor):
    x = x.contiguous()
    y = torch.empty_like(x)
    _ck(lib().muvo_divide_scalar(_f(x), _f(y), _i64(x.numel()), _fl(divisor), _st()))
    return y


def resize_bilinear(x, oh, ow):
    """x (..., H, W) -> (..., oh, ow), bilinear align_corners=False without antialias."""
    x = x.contiguous()
    h, w = x.shape[-2:]
    y = torch.empty(*x.shape[:-2], oh, ow, device=x.device, dtype=torch.float32)
    _ck(lib().muvo_resize_bilinear(_f(x), _f(y), _i64(x.numel() // (h * w)), h, w, oh, ow, _st()))
    return y


class _ResizeBilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        ctx.in_hw = tuple(x.shape[-2:])
        return resize_bilinear(x, oh, ow)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        h, w = ctx.in_hw
        oh, ow = g.shape[-2:]
        dx = torch.empty(*g.shape[:-2], h, w, device=g.device, dtype=torch.float32)
        _ck(lib().muvo_resize_bilinear_bwd(_f(g), _f(dx), _i64(g.numel() // (oh * ow)), h, w, oh, ow, _st()))
        return dx, None, None


def interpolate_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) with autograd (common.py:96)."""
    return _ResizeBilinearFn.apply(x, int(size[0]), int(size[1]))


class _SoftmaxChannelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, C = x.shape[:2]
        y = torch.empty_like(x)
        _ck(lib().muvo_softmax_channel_fwd(_f(x), _f(y), B, C, _i64(x[0, 0].numel()), _st()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        B, C = y.shape[:2]
        dx = torch.empty_like(y)
        _ck(lib().muvo_softmax_channel_bwd(_f(y), _f(g.contiguous()), _f(dx), B, C, _i64(y[0, 0].numel()), _st()))
        return dx


def instance_labels(instance_label, sigma, ignore_index=255):
    """convert_instance_mask_to_center_and_offset_label (instance_utils.py:4-35): instance ids (b, s, 1, h, w) ->
    center (b, s, 1, h, w), offset (b, s, 2, h, w) float32."""
    b, s, _, h, w = instance_label.shape
    inst = instance_label.to(torch.uint8).contiguous()
    center = torch.empty(b, s, 1, h, w, device=inst.device, dtype=torch.float32)
    offset = torch.empty(b, s, 2, h, w, device=inst.device, dtype=torch.float32)
    scratch = torch.empty(b * s * 256 * 3, device=inst.device, dtype=torch.float64)
    _ck(lib().muvo_instance_labels(_p(inst), _i64(b * s), h, w, _fl(sigma), _fl(ignore_index), _p(scratch), _f(center), _f(offset), _st()))
    return center, offset


def softmax_channel(x):
    """x.softmax(dim=1) of an (B, C, ...) tensor (mile.py:509)."""
    return _SoftmaxChannelFn.apply(x)


def resize_nearest(x, out_sz):
    """nearest resize of the trailing len(out_sz) dims (2 or 3), float32 or uint8."""
    x = x.contiguous()
    nd = len(out_sz)
    in_sz = tuple(x.shape[-nd:])
    i3 = (1,) * (3 - nd) + in_sz
    o3 = (1,) * (3 - nd) + tuple(out_sz)
    y = torch.empty(*x.shape[:-nd], *out_sz, device=x.device, dtype=x.dtype)
    nc = x.numel() // (i3[0] * i3[1] * i3[2])
    fn = lib().muvo_resize_nearest_u8 if x.dtype == torch.uint8 else lib().muvo_resize_nearest_f32
    assert x.dtype in (torch.uint8, torch.float32)
    _ck(fn(_p(x), _p(y), _i64(nc), *i3, *o3, _st()))
    return y


# ================================================================================================ losses
def _grad_vector(gs, device):
    """the gradients of k scalar outputs of a loss Function (`terms=True`: returned as separate 0-d tensors, so that the caller's
    `[i]` is a tuple index and not k select_backward nodes of 2-3 launches each) as ONE (k,) device vector for the backward kernel;
    a term nobody differentiated counts as zero.  One launch (all terms usually receive the same upstream scalar)."""
    first = next((g for g in gs if g is not None), None)
    if first is None:
        return torch.zeros(len(gs), device=device, dtype=torch.float32)
    if all(g is first for g in gs):
        return first.reshape(1).expand(len(gs)).contiguous()
    zero = None
    parts = []
    for g in gs:
        if g is None:
            zero = torch.zeros((), device=device, dtype=torch.float32) if zero is None else zero
            g = zero
        parts.append(g.reshape(()))
    return torch.stack(parts)


def _as_terms(ctx, t, terms):
    if not terms:
        return t
    ctx.set_materialize_grads(False)
    return tuple(t[i] for i in range(t.numel()))


class SpatialLossFn(torch.autograd.Function):
    """weight * SpatialRegressionLoss(norm) over channel ranges of (B,S,C,H,W) tensors.

    `parts` = list of (c0, c1, norm, weight); returns one scalar per part (stacked 1-D tensor)."""

    @staticmethod
    def forward(ctx, pred, target, parts, ignore, mask=None, terms=False):
        # mask: optional explicit (B, S, 1, H, W) uint8 / bool pixel mask (SpatialRegressionLoss(..., instance_mask), losses.py:87-90)
        pred, target = pred.contiguous(), target.contiguous()
        b, s, c, h, w = pred.shape
        f, hw = b * s, h * w
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
            assert mask.numel() == f * hw
        losses = torch.empty(len(parts), device=pred.device, dtype=torch.float32)
        stats = torch.empty(len(parts), 2, device=pred.device, dtype=torch.float64)
        for i, (c0, c1, norm, weight) in enumerate(parts):
            _ck(lib().muvo_spatial_loss_masked_fwd(_f(pred), _f(target), _p(mask), _i64(f), c, _i64(hw), c0, c1, norm, _fl(ignore),
                                                   _fl(weight), C.c_void_p(stats.data_ptr() + 16 * i),
                                                   C.c_void_p(losses.data_ptr() + 4 * i), _st()))
        ctx.parts, ctx.ignore, ctx.dims = parts, ignore, (f, c, hw)
        ctx.save_for_backward(pred, target, stats, mask)
        return _as_terms(ctx, losses, terms)

    @staticmethod
    def backward(ctx, *gs):
        pred, target, stats, mask = ctx.saved_tensors
        f, c, hw = ctx.dims
        g = gs[0].contiguous() if (len(gs) == 1 and gs[0] is not None and gs[0].dim() == 1) else _grad_vector(gs, pred.device)
        covered = sum(c1 - c0 for c0, c1, _, _ in ctx.parts)
        dpred = torch.empty_like(pred) if covered == c else torch.zeros_like(pred)
        for i, (c0, c1, norm, weight) in enumerate(ctx.parts):
            _ck(lib().muvo_spatial_loss_masked_bwd(_f(pred), _f(target), _p(mask), _f(dpred), _i64(f), c, _i64(hw), c0, c1, norm,
                                                   _fl(ctx.ignore), _fl(weight), C.c_void_p(stats.data_ptr() + 16 * i),
                                                   C.c_void_p(g.data_ptr() + 4 * i), _st()))
        return dpred, None, None, None, None, None


def spatial_losses(pred, target, parts, ignore=255.0, mask=None, terms=False):
    """terms=True: a tuple of 0-d tensors (one per part) instead of the stacked 1-D tensor"""
    return SpatialLossFn.apply(pred, target, parts, ignore, mask, terms)


class VoxelLossFn(torch.autograd.Function):
    """(weight*CE mean, weight*SemScal, weight*GeoScal) for logits (B,S,C,X,Y,Z), labels u8 (B,S,1,X,Y,Z)."""

    @staticmethod
    def forward(ctx, logits, target, weight, class_w, terms=False):
        logits, target = logits.contiguous(), target.contiguous()
        b, s, c = logits.shape[:3]
        v = logits.numel() // (b * s * c)
        L = lib()
        stats = torch.empty(L.muvo_voxel_loss_stats_doubles(c), device=logits.device, dtype=torch.float64)
        coef = torch.empty(L.muvo_voxel_loss_coef_floats(c), device=logits.device, dtype=torch.float32)
        loss3 = torch.empty(3, device=logits.device, dtype=torch.float32)
        _ck(L.muvo_voxel_loss_fwd(_f(logits), _p(target), _i64(b * s), c, _i64(v), _f(class_w), _fl(weight), _p(stats),
                                  _f(coef), _f(loss3), _st()))
        ctx.dims, ctx.weight, ctx.class_w = (b * s, c, v), weight, class_w
        ctx.save_for_backward(logits, target, coef)
        return _as_terms(ctx, loss3, terms)

    @staticmethod
    def backward(ctx, *gs):
        logits, target, coef = ctx.saved_tensors
        f, c, v = ctx.dims
        g = gs[0].contiguous() if (len(gs) == 1 and gs[0] is not None and gs[0].dim() == 1) else _grad_vector(gs, logits.device)
        dl = torch.empty_like(logits)
        _ck(lib().muvo_voxel_loss_bwd(_f(logits), _p(target), _f(dl), _i64(f), c, _i64(v), _f(ctx.class_w),
                                      _fl(ctx.weight), _f(coef), _f(g), _st()))
        return dl, None, None, None, None


def voxel_losses(logits, target, weight, class_w=None, terms=False):
    return VoxelLossFn.apply(logits, target, weight, class_w, terms)


class L1RowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, terms=False):
        pred, target = pred.contiguous(), target.contiguous()
        cols = pred.shape[-1]
        rows = pred.numel() // cols
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        _ck(lib().muvo_l1_rows_fwd(_f(pred), _f(target), _i64(rows), cols, _fl(weight), _f(loss), _st()))
        ctx.dims, ctx.weight = (rows, cols), weight
        ctx.save_for_backward(pred, target)
        return _as_terms(ctx, loss, terms)

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        rows, cols = ctx.dims
        dp = torch.empty_like(pred)
        g = torch.zeros(1, device=pred.device) if g is None else g.reshape(1)
        _ck(lib().muvo_l1_rows_bwd(_f(pred), _f(target), _f(dp), _i64(rows), cols, _fl(ctx.weight), _f(g.contiguous()),
                                   _st()))
        return dp, None, None, None


class _SegCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, class_w):
        logits = logits.contiguous()
        N, Cc = logits.shape[:2]
        HW = logits[0, 0].numel()
        loss = torch.empty(N, HW, device=logits.device, dtype=torch.float32)
        _ck(lib().muvo_seg_ce_fwd(_f(logits), _p(target), _f(class_w), _f(loss), _i64(N), Cc, _i64(HW), _st()))
        ctx.save_for_backward(logits, target, class_w)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target, class_w = ctx.saved_tensors
        N, Cc = logits.shape[:2]
        HW = logits[0, 0].numel()
        d = torch.empty_like(logits)
        _ck(lib().muvo_seg_ce_bwd(_f(logits), _p(target), _f(class_w), _f(g.contiguous()), _f(d), _i64(N), Cc, _i64(HW), _st()))
        return d, None, None


def seg_ce_pixel_loss(logits, target, class_w=None):
    """F.cross_entropy(logits (N, C, ...), target (N, ...), weight=class_w, reduction='none') flattened to (N, HW)."""
    t = target.reshape(target.shape[0], -1).to(torch.uint8).contiguous()
    return _SegCEFn.apply(logits.float(), t, class_w)


def l1_rows_loss(pred, target, weight, terms=False):
    return L1RowsFn.apply(pred, target, weight, terms)


class KLLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pm, ps, qm, qs, weight, alpha, terms=False):
        pm, ps, qm, qs = (t.contiguous() for t in (pm, ps, qm, qs))
        b, t, s = pm.shape
        loss = torch.empty(1, device=pm.device, dtype=torch.float32)
        _ck(lib().muvo_kl_loss_fwd(_f(pm), _f(ps), _f(qm), _f(qs), b, t, s, _fl(weight), _f(loss), _st()))
        ctx.args = (b, t, s, weight, alpha)
        ctx.save_for_backward(pm, ps, qm, qs)
        return _as_terms(ctx, loss, terms)

    @staticmethod
    def backward(ctx, g):
        pm, ps, qm, qs = ctx.saved_tensors
        b, t, s, weight, alpha = ctx.args
        g = torch.zeros(1, device=pm.device) if g is None else g.reshape(1)
        d = [torch.empty_like(pm) for _ in range(4)]
        _ck(lib().muvo_kl_loss_bwd(_f(pm), _f(ps), _f(qm), _f(qs), _f(d[0]), _f(d[1]), _f(d[2]), _f(d[3]), b, t, s,
                                   _fl(weight), _fl(alpha), _f(g.contiguous()), _st()))
        return d[0], d[1], d[2], d[3], None, None, None


def kl_loss(pm, ps, qm, qs, weight, alpha, terms=False):
    return KLLossFn.apply(pm, ps, qm, qs, weight, alpha, terms)


# ================================================================================================ optimiser
def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _ck(lib().muvo_adamw_step(_f(p), _f(g), _f(m), _f(v), _i64(p.numel()), _fl(lr), _fl(beta1), _fl(beta2), _fl(eps),
                              _fl(weight_decay), int(step), _fl(grad_scale), _st()))


# ================================================================================================ time stack / unstack
class StackTimeFn(torch.autograd.Function):
    """list of s tensors (b, D) -> (b, s, D) via strided 2-D copies."""

    @staticmethod
    def forward(ctx, *xs):
        s = len(xs)
        b, d = xs[0].shape
        y = torch.empty(b, s, d, device=xs[0].device, dtype=torch.float32)
        for t, x in enumerate(xs):
            x = x.contiguous()
            _ck(lib().muvo_copy2d(_f(x), C.c_void_p(y.data_ptr() + 4 * t * d), _i64(b), _i64(d), _i64(d), _i64(s * d), 0,
                                  _st()))
        ctx.dims = (b, s, d)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, s, d = ctx.dims
        dy = dy.contiguous()
        outs = []
        for t in range(s):
            if ctx.needs_input_grad[t]:
                g = torch.empty(b, d, device=dy.device, dtype=torch.float32)
                _ck(lib().muvo_copy2d(C.c_void_p(dy.data_ptr() + 4 * t * d), _f(g), _i64(b), _i64(d), _i64(s * d), _i64(d),
                                      0, _st()))
                outs.append(g)
            else:
                outs.append(None)
        return tuple(outs)


def stack_time(xs):
    return StackTimeFn.apply(*xs)


class UnstackTimeFn(torch.autograd.Function):
    """(b, s, D) -> tuple of s contiguous (b, D) tensors."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        b, s, d = x.shape
        outs = []
        for t in range(s):
            g = torch.empty(b, d, device=x.device, dtype=torch.float32)
            _ck(lib().muvo_copy2d(C.c_void_p(x.data_ptr() + 4 * t * d), _f(g), _i64(b), _i64(d), _i64(s * d), _i64(d), 0,
                                  _st()))
            outs.append(g)
        ctx.dims = (b, s, d)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        b, s, d = ctx.dims
        dev = next(g.device for g in gs if g is not None)
        dx = torch.zeros(b, s, d, device=dev, dtype=torch.float32)
        for t, g in enumerate(gs):
            if g is None:
                continue
            g = g.contiguous()
            _ck(lib().muvo_copy2d(_f(g), C.c_void_p(dx.data_ptr() + 4 * t * d), _i64(b), _i64(d), _i64(d), _i64(s * d), 0,
                                  _st()))
        return dx


def unstack_time(x):
    return UnstackTimeFn.apply(x)


class SegmentMarkFn(torch.autograd.Function):
    """Identity whose backward calls `cb(name)` first: placed on the input of a sub-network it tells the data-parallel
    reducer that backward has issued every kernel of that sub-network (muvo_amd/parallel.py)."""

    @staticmethod
    def forward(ctx, x, cb, name):
        ctx.cb, ctx.name = cb, name
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.cb(ctx.name)
        return g, None, None


def segment_mark(x, cb, name):
    return SegmentMarkFn.apply(x, cb, name)


class SumScalarsFn(torch.autograd.Function):
    """total = sum of 0-d tensors (trainer.py:511-513 loss_reducing)."""

    @staticmethod
    def forward(ctx, *vals):
        out = torch.zeros(1, device=vals[0].device, dtype=torch.float32)
        for v in vals:
            _ck(lib().muvo_copy2d(C.c_void_p(v.data_ptr()), _f(out), _i64(1), _i64(1), _i64(1), _i64(1), 1, _st()))
        ctx.n = len(vals)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return tuple(g for _ in range(ctx.n))


def sum_scalars(vals):
    return SumScalarsFn.apply(*vals)


# ------------------------------------------------------------------------------------------------
# one-GPU stand-in for a gradient all-reduce among `peers` GPUs (include/muvo_hip.h: muvo_fake_allreduce)
_FAKE_AR = {}


def fake_allreduce(buf, peers=8, workgroups=None, ms_per_100mb=None):
    """Runs on the current stream; leaves buf bit-identical.  Rate: 1 ms per 100 MB (an 8-GPU xGMI ring at ~175 GB/s of bus
    bandwidth: 2 (G-1)/G x 100 MB / 175 GB/s; RCCL reaches more on large messages, so this is the pessimistic side;
    MUVO_DP_FAKE_MS_PER_100MB overrides).  The sleep count that gives this rate with
    `workgroups` resident workgroups (RCCL-like: MUVO_DP_FAKE_WGS, default 32) is calibrated once per device on a scratch buffer."""
    dev = buf.device
    wgs = int(workgroups or os.environ.get('MUVO_DP_FAKE_WGS', '32'))
    target = float(ms_per_100mb or os.environ.get('MUVO_DP_FAKE_MS_PER_100MB', '1.0'))
    key = (dev.index, wgs, round(target, 4))
    sleep = _FAKE_AR.get(key)
    L = lib()
    if sleep is None:
        probe = torch.zeros(16 * 2 ** 20, device=dev, dtype=torch.float32)        # 64 MB
        ms = {}
        for sl in (0, 64):
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _ck(L.muvo_fake_allreduce(_f(probe), _i64(probe.numel()), wgs, sl, _st()))
                e1.record()
                e1.synchronize()
                ms[sl] = e0.elapsed_time(e1) * 100.0 / 64.0            # per 100 MB
        per = max((ms[64] - ms[0]) / 64.0, 1e-6)
        sleep = int(max(0, round((target - ms[0]) / per)))
        _FAKE_AR[key] = sleep
        _FAKE_AR[('calibration',) + key] = dict(ms_per_100mb_sleep0=ms[0], ms_per_100mb_sleep64=ms[64], sleep=sleep, target=target, workgroups=wgs)
    n = buf.numel() & ~3
    if n:
        _ck(L.muvo_fake_allreduce(_f(buf), _i64(n), wgs, sleep, _st()))


def fake_allreduce_calibration():
    return [v for k, v in _FAKE_AR.items() if k and k[0] == 'calibration']
