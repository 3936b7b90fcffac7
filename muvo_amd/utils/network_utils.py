"""(b, s, ...) <-> (b*s, ...) views (muvo/utils/network_utils.py:30-78)."""
import torch


def pack_sequence_dim(x):
    if isinstance(x, torch.Tensor):
        b, s = x.shape[:2]
        return x.reshape(b * s, *x.shape[2:])
    if isinstance(x, list):
        return [pack_sequence_dim(e) for e in x]
    return {k: pack_sequence_dim(v) for k, v in x.items()}


def unpack_sequence_dim(x, b, s):
    if isinstance(x, torch.Tensor):
        return x.view(b, s, *x.shape[1:])
    if isinstance(x, list):
        return [unpack_sequence_dim(e, b, s) for e in x]
    return {k: unpack_sequence_dim(v, b, s) for k, v in x.items()}


def remove_past(x, receptive_field):
    if isinstance(x, torch.Tensor):
        return x[:, (receptive_field - 1):].contiguous()
    return {k: remove_past(v, receptive_field) for k, v in x.items()}
