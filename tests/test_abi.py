"""CPU: the C-ABI shared library loads and exports every symbol include/muvo_hip.h declares (no compute calls)."""
import os
import re

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'muvo_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(muvo_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_header_symbols():
    import ctypes
    from muvo_amd import build, ops
    lib_path = build.build(verbose=False)
    assert os.path.exists(lib_path)
    L = ctypes.CDLL(lib_path)
    syms = _header_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(ops.EXPORTS) == syms, set(ops.EXPORTS) ^ set(syms)
    L.muvo_abi_version.restype = ctypes.c_int
    assert L.muvo_abi_version() == 1


def test_argument_validation_without_gpu():
    """Entry points validate arguments before touching the device: bad descriptors return MUVO_ERR_INVALID_ARG."""
    import ctypes as C
    from muvo_amd import ops
    L = ops.lib()
    d = ops.ConvDesc(2, 0, 1, 3, 8, (C.c_int32 * 3)(1, 8, 8), (C.c_int32 * 3)(1, 7, 8), (C.c_int32 * 3)(1, 3, 3),
                     (C.c_int32 * 3)(1, 1, 1), (C.c_int32 * 3)(0, 1, 1), (C.c_int32 * 3)(1, 1, 1))
    a, b = C.c_int64(0), C.c_int64(0)
    assert L.muvo_conv_pack_sizes(C.byref(d), C.byref(a), C.byref(b)) == -1
    assert b'out_sz' in L.muvo_last_error()
    d.out_sz[1] = 8
    assert L.muvo_conv_pack_sizes(C.byref(d), C.byref(a), C.byref(b)) == 0
    assert a.value == 48 * 32 and b.value > 0  # C=3 -> Cp=4, 9 taps -> K=36 -> Kp=48 rows; M=8 -> Mp=32
