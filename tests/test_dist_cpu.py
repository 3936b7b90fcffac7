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
    assert [n for n, _, _ in store.segment_ranges] == ['rgb_decoder', 'rssm', 'fusion', 'image_branch']
    for step in range(2):
        red.begin_step()
        store.zero_grad()
        for n, p in m.named_parameters():
            if n in used:
                if step == 1 and n == 'encoder.weight':
                    # a FOREIGN gradient tensor (what AccumulateGrad installs after `p.grad = None`): the reducer must copy it
                    # into the flat slot before the segment is exchanged, or it is never reduced
                    p.grad = torch.full_like(p, float(rank + 1) * (step + 1))
                else:
                    p.grad.add_(float(rank + 1) * (step + 1))
        if overlap:
            red.segment_done('policy')   # no such segment in the toy: everything before it in layout order = 'rgb_decoder'
            red.segment_done('fusion')   # implies 'rssm'
            assert red.launch_log == [('rgb_decoder', True), ('rssm', True), ('fusion', True)]
        red.finish()
        assert [n for n, _ in red.launch_log] == ['rgb_decoder', 'rssm', 'fusion', 'image_branch']
        expect = float(sum(range(1, world + 1))) * (step + 1)
        assert torch.all(store.flat_grad[~pad] == expect), (store.flat_grad, expect)
        assert torch.all(store.flat_grad[pad] == 0) and torch.all(store.flat_param[pad] == 0)
        assert m.encoder_layer.weight.grad is None
        assert m.encoder.weight.grad.data_ptr() == store.flat_grad.data_ptr() + 4 * store.offsets['encoder.weight'][0]
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


def test_optimizer_state_dict_format_cpu():
    """FusedAdamW.state_dict()/load_state_dict() speak torch.optim.AdamW's format (host logic only: no kernel runs)."""
    sys.path.insert(0, ROOT)
    from muvo_amd.optim import FusedAdamW
    from muvo_amd.param_store import ParamStore
    torch.manual_seed(0)
    m = _Toy()
    store = ParamStore(m)
    opt = FusedAdamW(store, lr=1e-3, weight_decay=0.01)
    # groups in the reference's order: 1-D tensors (biases) -> no decay, matrices -> decay; the unused module is listed too
    assert [len(g['params']) for g in opt.param_groups] == [5, 5]
    assert opt.state_dict()['state'] == {}
    opt._step = 3
    store.exp_avg.uniform_(-1, 1)
    store.exp_avg_sq.uniform_(0, 1)
    sd = opt.state_dict()
    assert len(sd['state']) == 8 and all(float(v['step']) == 3.0 for v in sd['state'].values())   # encoder_layer.*: no state
    twin = [torch.nn.Parameter(torch.zeros(p.shape)) for g in opt.param_groups for p in g['params']]
    ref = torch.optim.AdamW([{'params': twin[:5], 'weight_decay': 0.0}, {'params': twin[5:], 'weight_decay': 0.01}], lr=1e-3)
    ref.load_state_dict(sd)                        # torch accepts it ...
    back = ref.state_dict()                        # ... and what torch writes loads back
    m2 = _Toy()
    store2 = ParamStore(m2)
    opt2 = FusedAdamW(store2, lr=5e-4, weight_decay=0.01)
    opt2.load_state_dict(back)
    assert opt2._step == 3 and opt2.param_groups[0]['lr'] == 1e-3
    pad = torch.ones(store.numel, dtype=torch.bool)
    for o, k in store.offsets.values():
        pad[o:o + k] = False
    assert torch.equal(store2.exp_avg[~pad], store.exp_avg[~pad]) and torch.equal(store2.exp_avg_sq[~pad], store.exp_avg_sq[~pad])
    # gradients re-bind to the flat buffer after nn.Module.zero_grad(set_to_none=True); foreign gradient tensors are copied in
    m.zero_grad(set_to_none=True)
    assert m.rssm.weight.grad is None
    m.rssm.weight.grad = torch.full_like(m.rssm.weight, 2.0)
    store.flat_grad.fill_(7.0)
    store.settle_grads()
    o, k = store.offsets['rssm.weight']
    assert torch.all(store.flat_grad[o:o + k] == 2.0) and m.rssm.weight.grad.data_ptr() == store.flat_grad.data_ptr() + 4 * o
    o, k = store.offsets['encoder.bias']
    assert torch.all(store.flat_grad[o:o + k] == 0.0)      # got no gradient since the reset: treated as zero


def test_zero_grad_discards_foreign_gradients():
    """ParamStore.zero_grad(): a gradient tensor that is not the flat view is dropped, not copied back into the zeroed slot
    (a skipped optimizer step followed by zero_grad must not leave the old gradient behind); settle_grads() - the optimizer's
    and the reducer's entry - copies foreign tensors in and zeroes the slots of parameters without a gradient."""
    sys.path.insert(0, ROOT)
    from muvo_amd.param_store import ParamStore
    torch.manual_seed(0)
    m = _Toy()
    store = ParamStore(m)
    p = m.encoder.weight
    o, k = store.offsets['encoder.weight']
    p.grad = torch.full_like(p, 2.0)
    store.zero_grad()
    assert torch.all(store.flat_grad == 0)
    assert p.grad.data_ptr() == store.flat_grad.data_ptr() + 4 * o
    p.grad = torch.full_like(p, 3.0)
    m.encoder.bias.grad = None
    store.flat_grad[store.offsets['encoder.bias'][0]] = 7.0          # stale value of an earlier step
    store.settle_grads([p, m.encoder.bias])
    assert torch.all(store.flat_grad[o:o + k] == 3.0)
    assert torch.all(m.encoder.bias.grad == 0) and p.grad.data_ptr() == store.flat_grad.data_ptr() + 4 * o
