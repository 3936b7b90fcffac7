"""CPU, world_size 2 over gloo: the segmented gradient reducer sums every segment exactly once (with and without
overlap hooks) and the flat ParamStore layout covers every used parameter."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.rgb_decoder = torch.nn.Linear(4, 3)
        self.rssm = torch.nn.Linear(3, 3)
        self.transformer_encoder = torch.nn.Linear(3, 2)
        self.encoder = torch.nn.Linear(2, 2)
        self.encoder_layer = torch.nn.Linear(2, 2)  # registered-but-unused (never reduced, never updated)


def _worker(rank, world, port, overlap, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from muvo_amd.parallel import SegmentedGradReducer
    from muvo_amd.param_store import ParamStore
    torch.manual_seed(0)
    m = _Toy()
    store = ParamStore(m)
    red = SegmentedGradReducer(store, overlap=overlap)
    used = [n for n, _ in m.named_parameters() if not n.startswith('encoder_layer')]
    assert sorted(store.offsets) == sorted(used)
    n_el = sum(p.numel() for n, p in m.named_parameters() if n in used)
    assert n_el <= store.numel <= n_el + 3 * len(used)                        # tensors start on 16-byte boundaries
    assert all(o % 4 == 0 for o, _ in store.offsets.values())
    pad = torch.ones(store.numel, dtype=torch.bool)
    for o, k in store.offsets.values():
        pad[o:o + k] = False
    assert [n for n, _, _ in store.segment_ranges] == ['decoders', 'rssm', 'fusion', 'encoders']
    for step in range(2):
        red.begin_step()
        store.zero_grad()
        for n, p in m.named_parameters():
            if n in used:
                p.grad.add_(float(rank + 1) * (step + 1))
        if overlap:
            red.segment_done('decoders')
            red.segment_done('fusion')   # implies 'rssm'
        red.finish()
        expect = float(sum(range(1, world + 1))) * (step + 1)
        assert torch.all(store.flat_grad[~pad] == expect), (store.flat_grad, expect)
        assert torch.all(store.flat_grad[pad] == 0) and torch.all(store.flat_param[pad] == 0)
        assert m.encoder_layer.weight.grad is None
    assert red.grad_scale == 1.0 / world
    q.put(rank)
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap', [False, True])
def test_segmented_reducer_gloo(overlap):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (1 if overlap else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [0, 1]
