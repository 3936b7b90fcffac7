"""MI355X-native MUVO world-model training step."""
import os

# The step runs independent sub-networks on a few HIP streams (muvo_amd/ops.py: side_stream).  With the runtime's default of four
# hardware queues per process, streams of other libraries (RCCL) push them onto shared queues in an order nobody controls; eight
# queues give every stream its own.  Read by the HIP runtime when it initialises, i.e. at the first GPU call - importing this
# package before touching the GPU is enough.  An explicit setting in the environment wins.
HW_QUEUES_PRESET = 'GPU_MAX_HW_QUEUES' in os.environ       # set by the caller's shell (the only way under rocprofv3, see below)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def hw_queue_state():
    """How sure are we that the HIP runtime saw GPU_MAX_HW_QUEUES?  'preset': it was in the environment before this package was
    imported (the shell's env prefix: under rocprofv3 the profiler's preloaded library initialises the GPU before Python
    starts, so only that form reaches the runtime); 'set-at-import': this package set it and torch had not initialised the GPU
    yet; 'too-late': the GPU was already initialised when this package was imported and the variable was not preset - the
    runtime runs on its default of four hardware queues, the stream plan of muvo_amd/ops.py assumes eight."""
    return _HWQ_STATE


def _hwq_state():
    if HW_QUEUES_PRESET:
        return 'preset'
    try:
        import sys
        torch = sys.modules.get('torch')
        if torch is not None and torch.cuda.is_initialized():
            import warnings
            warnings.warn('muvo_amd: the GPU was initialised before muvo_amd was imported and GPU_MAX_HW_QUEUES was not set in the '
                          'environment: the HIP runtime keeps its default of four hardware queues (set GPU_MAX_HW_QUEUES=8 in the shell)')
            return 'too-late'
    except Exception:
        pass
    if os.environ.get('ROCPROFILER_REGISTER_FORCE_LOAD') or 'rocprofiler' in os.environ.get('LD_PRELOAD', ''):
        return 'too-late (profiler preload initialised the GPU first)'
    return 'set-at-import'


_HWQ_STATE = _hwq_state()
