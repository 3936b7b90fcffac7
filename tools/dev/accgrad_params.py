"""bench.py's stderr carried autograd's warning "The AccumulateGrad node's stream does not match the stream of the node that produced
the incoming gradient".  This probe answers: (1) which parameters receive a DEFINED gradient tensor through an AccumulateGrad node
(all others get theirs written by kernels into the flat gradient buffer; their AccumulateGrad node runs on an undefined input and
does nothing), and on which stream; (2) in which kind of step the warning is raised: the timed configuration (side streams on), or
the untimed per-class instrumentation steps bench.py runs afterwards with the side streams OFF while the autograd graph of the
previous step - built on the side streams - is still alive.     python tools/dev/accgrad_params.py   (GPU box)"""
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import muvo_amd  # noqa
import torch
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch
from muvo_amd.trainer import WorldModelTrainer

dev = torch.device('cuda', 0)
tr = WorldModelTrainer(base_1d_cfg(RECEPTIVE_FIELD=6, FUTURE_HORIZON=4, BATCHSIZE=2, STEPS=100000).convert_to_dict(), device=dev)
tr.train()
opt = tr.configure_optimizers()[0][0]
hits = {}
main = torch.cuda.current_stream(dev)
for n, p in tr.model.named_parameters():
    def hook(g, n=n):
        cur = torch.cuda.current_stream(dev)
        hits.setdefault(n, []).append('main' if cur == main else hex(cur.cuda_stream))
    p.register_hook(hook)
batches = [make_batch(2, 10, seed=1234 + k, device=dev) for k in range(2)]


def step(i):
    opt.zero_grad()
    loss = tr.training_step(dict(batches[i % 2]), i)
    loss.backward()
    tr.on_after_backward()
    opt.step()
    return loss


def count_warnings(fn):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out = fn()
        torch.cuda.synchronize()
    return out, sum('AccumulateGrad' in str(x.message) for x in w)


keep, n_on = count_warnings(lambda: [step(i) for i in range(3)][-1])
print('side streams:', {k[0]: hex(v.cuda_stream) for k, v in ops._side_streams.items()})
print(f'{len(hits)} of {sum(1 for _ in tr.model.parameters())} parameters have an AccumulateGrad node that runs (tensor inputs of autograd Functions; the node runs\n'
      f'even when the Function returns None for the parameter - the kernels wrote the gradient - and then does nothing), 3 steps:')
for n, s in hits.items():
    p = dict(tr.model.named_parameters())[n]
    print(f'  {n:50s} {str(tuple(p.shape)):18s} hook ran under streams {sorted(set(s))}; p.grad is a view of the flat buffer: '
          f'{p.grad.data_ptr() == p._muvo_flat_grad.data_ptr()}')
print(f'AccumulateGrad stream-mismatch warnings in 3 steps with the side streams ON (the timed configuration): {n_on}')
ops.STREAMS = ops.WGRAD_STREAM = False
_, n_off_alive = count_warnings(lambda: step(3))          # `keep` (the previous loss) holds the previous graph alive
print(f'... in one step with the side streams OFF while the previous graph is alive (bench.py\'s per-class instrumentation steps): {n_off_alive}')
del keep
_, n_off = count_warnings(lambda: step(4))
print(f'... in the next side-streams-OFF step: {n_off}')
