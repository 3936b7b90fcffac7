"""-m gpu: the plain-loop launcher (muvo_amd/train.py; reference train.py:51-115 under Lightning): loss logging, the checkpoint
callback's file convention (train.py:31-48: Lightning's file name, written into the CWD), resume, gradient accumulation."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(**kw):
    from muvo_amd.config import base_1d_cfg
    base = dict(RECEPTIVE_FIELD=2, FUTURE_HORIZON=0, STEPS=4, BATCHSIZE=1, VAL_CHECK_INTERVAL=2, LOGGING_INTERVAL=1)
    base.update(kw)
    return base_1d_cfg(**base)


def test_fit_checkpoint_resume(dev, tmp_path, monkeypatch):
    from muvo_amd import train
    monkeypatch.chdir(tmp_path)
    lines = []
    module, hist = train.fit(_cfg(), dev, log=lines.append)
    assert [h['step'] for h in hist] == [1, 2, 3, 4]
    assert sum(k.startswith('train_') for k in hist[0]) == 21 and '-global_step' in hist[0]
    assert sorted(f for f in os.listdir('.') if f.endswith('.ckpt')) == ['epoch=0-step=2.ckpt', 'epoch=0-step=4.ckpt']
    assert any(l.startswith('checkpoint epoch=0-step=2.ckpt') for l in lines)
    ck = torch.load('epoch=0-step=2.ckpt', map_location='cpu', weights_only=False)
    assert ck['global_step'] == 2 and len(ck['optimizer_states'][0]['state']) == 440
    names = set(ck['state_dict'])
    assert {'preprocess.image_mean', 'preprocess.image_std', 'model.type_embedding', 'model.rssm.recurrent_model.weight_hh'} <= names
    assert sum(n.startswith('model.') for n in names) == 680
    # the reference's own weight loader accepts the file (trainer.py:202-211: PRETRAINED.PATH, 'model.' prefix stripped, strict)
    from muvo_amd.trainer import WorldModelTrainer
    tr = WorldModelTrainer(_cfg().convert_to_dict(), pretrained_path=os.path.abspath('epoch=0-step=2.ckpt'), device=dev)
    assert torch.equal(tr.model.type_embedding.cpu(), ck['state_dict']['model.type_embedding'])
    # resume at step 2: steps 3 and 4 reproduce the uninterrupted run (same seeds per step; float-atomic noise only)
    _, hist2 = train.fit(_cfg(), dev, resume='epoch=0-step=2.ckpt', log=lines.append)
    assert [h['step'] for h in hist2] == [3, 4]
    for a, b in zip(hist[2:], hist2):
        assert a['lr'] == b['lr']
        for k in a:
            if k.startswith('train_'):
                assert abs(a[k] - b[k]) <= 2e-3 * abs(a[k]), (a['step'], k, a[k], b[k])
    # the losses move: one optimisation is happening
    tot = [sum(v for k, v in h.items() if k.startswith('train_')) for h in hist]
    assert all(torch.isfinite(torch.tensor(tot))) and len(set(tot)) == 4


def test_gradient_accumulation_equals_big_batch_gradient(dev, tmp_path, monkeypatch):
    """ACCUMULATE_GRAD_BATCHES = 2 (reference default schedule: batch 1 x 16 accumulation, muvo.yml:10-19; Lightning divides the
    loss by the number of micro-batches): the accumulated flat gradient equals the mean of the two micro-batch gradients.  All
    randomness is pinned (constant RSSM noise, no prior-sample coin, no augmentation draw fires, dropout off)."""
    from muvo_amd import ops, train
    from muvo_amd.data.synthetic import make_batch
    monkeypatch.chdir(tmp_path)
    def const(value):
        def f(*shape, **kw):
            shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
            return torch.full(shape, value, **kw)
        return f
    monkeypatch.setattr(torch, 'randn', const(0.3))
    monkeypatch.setattr(torch, 'rand', const(0.9))

    def no_dropout(m):
        for layer in m.model.transformer_encoder.layers:
            layer.p = 0.0

    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_F32, min_gflop=-1.0)
    try:
        batches = [make_batch(1, 2, seed=4000 + k, device=dev) for k in range(2)]
        cfg = _cfg(STEPS=1000, VAL_CHECK_INTERVAL=0)
        cfg.OPTIMIZER.ACCUMULATE_GRAD_BATCHES = 2
        module, _ = train.fit(cfg, dev, steps=1, log=lambda s: None, batch_fn=lambda i: dict(batches[i]), setup=no_dropout)
        acc = module.store.flat_grad.double().clone()
        single = []
        for k in range(2):
            m1, _ = train.fit(_cfg(STEPS=1000, VAL_CHECK_INTERVAL=0), dev, steps=1, log=lambda s: None,
                              batch_fn=lambda i, _k=k: dict(batches[_k]), setup=no_dropout)
            single.append(m1.store.flat_grad.double().clone())
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)
    want = 0.5 * (single[0] + single[1])
    assert float(want.norm()) > 0
    assert float((acc - want).norm() / want.norm()) < 5e-3
    assert float((acc - single[0]).norm() / want.norm()) > 0.1          # (the two micro-batches really differ)
