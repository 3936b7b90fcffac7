"""CPU: the oracle restatement (oracle/muvo_ref.py, torch fp32) against the golden vectors produced by the REAL
reference (oracle/refimport/make_golden.py, run in the build container): state_dict names/shapes, the 21 losses,
output statistics and gradient norms on the seeded b=1, s=2 batch with deterministic weights."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def oracle_run():
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx = json.load(open(os.path.join(GOLD, 'base1d_b1s2.json')))
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    model = R.MileRef()
    detinit.fill_state_dict_(model)
    model.train()
    model.set_dropout(0.0)
    eps, use_prior = make_noise(fx['b'], fx['s'], seed=fx['seed'])
    batch = make_batch(fx['b'], fx['s'], seed=fx['seed'])
    total, losses, out, pbatch = R.training_step(model, batch, eps, use_prior)
    total.backward()
    return fx, model, total, losses, out, pbatch


def test_state_dict_matches_reference():
    from oracle import muvo_ref as R
    spec = json.load(open(os.path.join(GOLD, 'state_dict_spec.json')))
    m = R.MileRef()
    sd = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert sd == spec['state_dict']
    assert [n for n, _ in m.named_parameters()] == spec['parameters']
    assert len(sd) == 680 and len(spec['parameters']) == 452


def test_product_model_state_dict_matches_reference():
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.models.mile import Mile
    spec = json.load(open(os.path.join(GOLD, 'state_dict_spec.json')))
    m = Mile(base_1d_cfg())
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == spec['state_dict']
    assert [n for n, _ in m.named_parameters()] == spec['parameters']


def test_oracle_losses_match_reference(oracle_run):
    fx, _, total, losses, _, _ = oracle_run
    g = fx['steps'][0]
    assert set(losses) == set(g['losses']) and len(losses) == 21
    for k, v in g['losses'].items():
        assert abs(float(losses[k]) - v) <= 2e-5 * max(abs(v), 1e-6), (k, float(losses[k]), v)
    assert abs(float(total) - g['total']) < 2e-5 * g['total']


def test_oracle_outputs_match_reference(oracle_run):
    fx, _, _, _, out, pbatch = oracle_run
    smp = np.load(os.path.join(GOLD, 'base1d_b1s2_samples.npz'))
    for k, st in fx['steps'][0]['outputs'].items():
        if k.startswith('batch.'):
            t = pbatch[k[6:]]
        elif '.' in k:
            grp, name = k.split('.')
            t = out[grp][name]
        else:
            t = out[k]
        assert list(t.shape) == st['shape'], k
        f = t.detach().float().contiguous().view(-1)
        ref = torch.from_numpy(smp[('' if k.startswith('batch.') else 'out.') + k])
        got = f[::st['stride']][:ref.numel()]
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5 * max(1.0, st['absmean'])), k


def test_oracle_gradients_match_reference(oracle_run):
    fx, model, *_ = oracle_run
    g = fx['steps'][0]['grad_l2']
    for n, p in model.named_parameters():
        if g[n] is None:
            assert p.grad is None, n
        else:
            got = float(p.grad.double().pow(2).sum().sqrt())
            assert abs(got - g[n]) <= 1e-3 * g[n] + 1e-9, (n, got, g[n])


def test_oracle_losses_b2s4_prior_branch():
    """The second reference fixture (batch 2 x 4 frames, the RSSM continues from the PRIOR sample after t = 2): the oracle's 21
    losses, forward only (the generator already compared losses, outputs, gradients and two optimizer steps of the oracle with
    the reference on this batch: fixture key oracle_vs_reference)."""
    from muvo_amd.data.synthetic import make_batch, make_noise
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx = json.load(open(os.path.join(GOLD, 'base1d_b2s4.json')))
    assert fx['use_prior'] == [False, False, True, False]
    assert fx['steps'][0]['oracle_vs_reference']['max_rel_loss_dev'] < 1e-6
    assert fx['steps'][0]['oracle_vs_reference']['max_rel_grad_dev'] < 1e-4
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    model = R.MileRef()
    detinit.fill_state_dict_(model)
    model.train()
    model.set_dropout(0.0)
    eps, use_prior = make_noise(fx['b'], fx['s'], seed=fx['seed'])
    with torch.no_grad():
        total, losses, _, _ = R.training_step(model, make_batch(fx['b'], fx['s'], seed=fx['seed']), eps, use_prior)
    g = fx['steps'][0]
    for k, v in g['losses'].items():
        assert abs(float(losses[k]) - v) <= 2e-5 * max(abs(v), 1e-6), (k, float(losses[k]), v)
    assert abs(float(total) - g['total']) < 2e-5 * g['total']
