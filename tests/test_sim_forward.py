"""Closed-loop inference entry points (`sim_run.py:74-75` calls `model.preprocess(batch)` then
`model.model.sim_forward(...)`; `trainer.py:218-221` `deployment_forward`): three consecutive calls each of the REAL reference
(tests/golden/sim_b1.*, oracle/refimport/make_golden_sim.py; the latent memory advances on calls 1 and 3, call 2 only re-reads it;
call 3 of sim_forward dreams).  CPU: the oracle restatement.  GPU: Mile.sim_forward / WorldModelTrainer.deployment_forward on the
HIP kernels - latent state and actions 1e-3 of the tensor scale, decoder outputs and imagined outputs 2e-3."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
KEYS = ['rgb_1', 'lidar_reconstruction_1', 'voxel_1', 'voxel_4']


def _fixture():
    return json.load(open(os.path.join(GOLD, 'sim_b1.json'))), np.load(os.path.join(GOLD, 'sim_b1_samples.npz'))


def _check(tag, rec, smp, out, imag, tol_state, tol_out):
    for k in ('throttle_brake', 'steering', 'hidden_state', 'sample'):
        ref = torch.from_numpy(smp[f'{tag}.{k}'])
        got = out[k].detach().reshape(-1).float().cpu()
        assert (got - ref).abs().max().item() <= tol_state * max(ref.abs().max().item(), 1e-3), (tag, k)
    for name, src in (('out', out), ('imagine', imag)):
        for k in KEYS:
            st = rec.get(f'{name}.{k}')
            if st is None:
                continue
            t = src[k]
            assert list(t.shape) == st['shape'], (tag, name, k)
            f = t.detach().float().contiguous().view(-1)
            ref = torch.from_numpy(smp[f'{tag}.{name}.{k}'])
            got = f[::st['stride']][:ref.numel()].cpu()
            assert (got - ref).abs().max().item() <= tol_out * max(st['absmean'], ref.abs().max().item(), 1e-6), (tag, name, k)
    if imag:
        for k in ('throttle_brake', 'steering'):
            ref = torch.from_numpy(smp[f'{tag}.imagine.{k}'])
            got = imag[k].detach().reshape(-1).float().cpu()
            assert (got - ref).abs().max().item() <= tol_state * max(ref.abs().max().item(), 1e-3), (tag, 'imagine', k)


def test_oracle_sim_forward_matches_reference():
    from muvo_amd.data.synthetic import make_batch
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx, smp = _fixture()
    om = R.MileRef()
    detinit.fill_state_dict_(om)
    om.train()
    om.set_dropout(0.0)
    st = R.SimState()
    with torch.no_grad():
        for call in range(3):
            pb = R.preprocess(make_batch(fx['b'], fx['s'], seed=fx['seed'] + call), om.cfg)
            out, imag = R.sim_forward(om, st, pb, call == 2, fx['rf'], fx['stride_frames'])
            _check(f'sim{call}', fx['sim'][call], smp, out, imag, 2e-5, 2e-4)
    assert st.count == fx['stride_frames'] - 1


@pytest.mark.gpu
def test_hip_sim_and_deployment_forward_match_reference(dev):
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.data.synthetic import make_batch
    from muvo_amd.trainer import WorldModelTrainer
    from muvo_amd.utils import detinit
    fx, smp = _fixture()
    b, s, rf, seed = fx['b'], fx['s'], fx['rf'], fx['seed']
    tr = WorldModelTrainer(base_1d_cfg(RECEPTIVE_FIELD=rf, FUTURE_HORIZON=2, STEPS=100000).convert_to_dict(), device=dev)
    tr.train()                                   # sim_run.py:49-52: train() mode, Dropout modules in eval()
    tr.preprocess.augment = False
    detinit.fill_state_dict_(tr.model)
    for layer in tr.model.transformer_encoder.layers:
        layer.p = 0.0
    state0 = {k: v.clone() for k, v in tr.model.state_dict().items()}
    m = tr.model
    for call in range(3):
        batch = tr.preprocess(make_batch(b, s, seed=seed + call, device=dev))
        zeros = torch.zeros(b, s, m.cfg.MODEL.TRANSITION.STATE_DIM, device=dev)    # (call 2 does not cut the past: s - 1 imagined steps)
        out, imag = m.sim_forward(batch, is_dreaming=(call == 2), noise=zeros)
        _check(f'sim{call}', fx['sim'][call], smp, out, imag, 1e-3, 2e-3)
        assert m.count == (fx['stride_frames'] - 1 if call != 1 else 0)
    m.load_state_dict(state0)
    m.last_h = m.last_sample = m.last_action = None
    m.count = 0
    for call in range(3):
        raw = make_batch(b, s, seed=seed + 10 + call, device=dev)
        raw['action'] = torch.cat([raw['throttle_brake'], raw['steering']], -1)
        out = tr.deployment_forward(raw, is_dreaming=False)
        _check(f'dep{call}', fx['deploy'][call], smp, out, {}, 1e-3, 2e-3)


# ------------------------------------------------------------------------------------------------------------------
# Mile.forward(batch, deployment=True) (muvo/models/mile.py:404-489): the whole-sequence deployment variant, pinned by the REAL
# reference at batch 2 x 3 frames (tests/golden/deploy_fwd_b2s3.*, oracle/refimport/make_golden_deploy_fwd.py)
DKEYS = ['rgb_1', 'rgb_4', 'lidar_reconstruction_1', 'voxel_1', 'voxel_4', 'throttle_brake', 'steering']


def _deploy_fixture():
    return (json.load(open(os.path.join(GOLD, 'deploy_fwd_b2s3.json'))), np.load(os.path.join(GOLD, 'deploy_fwd_b2s3_samples.npz')))


def _check_deploy(fx, smp, out, tol):
    flat = {k: out[k] for k in DKEYS}
    for grp in ('prior', 'posterior'):
        for k, v in out[grp].items():
            flat[f'{grp}.{k}'] = v
    assert set(flat) == set(fx['outputs'])
    for k, st in fx['outputs'].items():
        t = flat[k]
        assert list(t.shape) == st['shape'] and t.shape[1] == 1, k          # remove_past: one time step left
        f = t.detach().float().contiguous().view(-1)
        ref = torch.from_numpy(smp['out.' + k])
        got = f[::st['stride']][:ref.numel()].cpu()
        assert (got - ref).abs().max().item() <= tol * max(st['absmean'], ref.abs().max().item(), 1e-6), k


def test_oracle_forward_deployment_matches_reference():
    from muvo_amd.data.synthetic import make_batch
    from muvo_amd.utils import detinit
    from oracle import muvo_ref as R
    fx, smp = _deploy_fixture()
    model = R.MileRef()
    detinit.fill_state_dict_(model)
    model.train()
    model.set_dropout(0.0)
    raw = make_batch(fx['b'], fx['s'], seed=fx['seed'])
    with torch.no_grad():
        pb = R.preprocess(raw, model.cfg)
        pb['action'] = torch.cat([raw['throttle_brake'], raw['steering']], -1)
        out = model.forward_deployment(pb)
    _check_deploy(fx, smp, out, 2e-5)


@pytest.mark.gpu
def test_hip_forward_deployment_matches_reference(dev):
    from muvo_amd.config import base_1d_cfg
    from muvo_amd.data.synthetic import make_batch
    from muvo_amd.trainer import WorldModelTrainer
    from muvo_amd.utils import detinit
    fx, smp = _deploy_fixture()
    b, s = fx['b'], fx['s']
    tr = WorldModelTrainer(base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000).convert_to_dict(), device=dev)
    tr.train()
    tr.preprocess.augment = False
    detinit.fill_state_dict_(tr.model)
    for layer in tr.model.transformer_encoder.layers:
        layer.p = 0.0
    raw = make_batch(b, s, seed=fx['seed'], device=dev)
    raw['action'] = torch.cat([raw['throttle_brake'], raw['steering']], -1)
    batch = tr.preprocess(raw)
    out, state_dict = tr.model(batch, deployment=True, use_prior=[False] * s)
    assert set(state_dict) == {'prior', 'posterior'} and not any(v.requires_grad for v in out.values() if torch.is_tensor(v))
    _check_deploy(fx, smp, out, 2e-3)
