"""Deterministic, platform-independent tensor generators.

Both the golden-fixture generator (which loads the values into the imported reference
model in the build container) and the product model / tests / bench on the GPU box call
these functions, so the 705 MB of weights never have to be committed: every value is a
pure function of (parameter name, element index).  Integer hashing only (numpy uint64),
so results are bit-identical on every host.
"""
import hashlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def name_key(name: str) -> int:
    return int.from_bytes(hashlib.sha256(name.encode()).digest()[:8], 'little')


def hash_u64(key: int, n: int, offset: int = 0) -> np.ndarray:
    with np.errstate(over='ignore'):
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        return _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + np.uint64(key & 0xFFFFFFFFFFFFFFFF))


def uniform_pm1(key: int, n: int) -> np.ndarray:
    """float32 uniform in [-1, 1), 24 random bits per value."""
    bits = (hash_u64(key, n) >> np.uint64(40)).astype(np.float32)
    return bits * np.float32(2.0 / (1 << 24)) - np.float32(1.0)


def uniform_01(key: int, n: int) -> np.ndarray:
    bits = (hash_u64(key, n) >> np.uint64(40)).astype(np.float32)
    return bits * np.float32(1.0 / (1 << 24))


def normal(key: int, n: int) -> np.ndarray:
    """Approximately N(0,1) float32 (sum of 4 uniforms, exact in float32 arithmetic order)."""
    acc = np.zeros(n, dtype=np.float32)
    for j in range(4):
        acc = acc + uniform_pm1(key + 0x1000 * (j + 1), n)
    return acc * np.float32(np.sqrt(3.0 / 4.0))


def det_tensor_for(name: str, shape, dtype=torch.float32) -> torch.Tensor:
    """Deterministic initial value for a state_dict entry, chosen by name/shape."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    if name.endswith('num_batches_tracked'):
        return torch.zeros(shape, dtype=torch.long)
    if name.endswith('running_mean'):
        return torch.zeros(shape, dtype=dtype)
    if name.endswith('running_var'):
        return torch.ones(shape, dtype=dtype)
    u = uniform_pm1(name_key(name), n)
    if len(shape) >= 2:
        fan = float(np.prod(shape[1:]))
        if 'trans_conv' in name or 'pre_transpose_conv' in name:
            # ConvTranspose2d weight is [in, out, kh, kw]; stride-2 layers see taps/4 per output
            taps = float(np.prod(shape[2:]))
            fan = shape[0] * max(taps / 4.0, 1.0) if 'pre_transpose_conv.0.' not in name else float(shape[0])
        v = u * np.float32(np.sqrt(3.0 / fan))
    elif name.endswith('weight'):
        v = np.float32(1.0) + np.float32(0.1) * u
    else:
        v = np.float32(0.05) * u
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def fill_state_dict_(module: torch.nn.Module) -> None:
    """In-place deterministic initialisation of every parameter and buffer of `module`."""
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            t.copy_(det_tensor_for(name, t.shape, t.dtype if t.is_floating_point() else torch.float32).to(t.dtype))
