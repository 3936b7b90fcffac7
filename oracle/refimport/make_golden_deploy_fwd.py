"""Golden fixture of Mile.forward(batch, deployment=True) (muvo/models/mile.py:404-489; runs ONLY in the build container): the
whole-sequence deployment variant of the REAL reference - encoder over b x s frames, RSSM with the recorded `batch['action']` and
the distribution means (use_sample=False), remove_past to the last time step, policy and decoders on that one state - at batch 2,
three frames, train() mode with the Dropout modules in eval() (as sim_run.py:49-52 sets a deployed model up).  Checks the oracle
restatement and writes tests/golden/deploy_fwd_b2s3.{json,npz}.   Usage: python oracle/refimport/make_golden_deploy_fwd.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as G  # noqa: E402
from muvo_amd.data.synthetic import make_batch  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402

KEYS = ['rgb_1', 'rgb_4', 'lidar_reconstruction_1', 'voxel_1', 'voxel_4', 'throttle_brake', 'steering']


def main():
    b, s, seed = 2, 3, 8642
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = G.import_reference()
    cfg = G.effective_cfg_dict(ref_config)
    cfg['RECEPTIVE_FIELD'], cfg['FUTURE_HORIZON'], cfg['STEPS'] = s, 0, 100000
    trainer = ref_trainer.WorldModelTrainer(cfg)
    trainer.train()
    for m in trainer.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    trainer.preprocess.eval()
    model = trainer.model
    detinit.fill_state_dict_(model)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    raw = make_batch(b, s, seed=seed)
    raw['action'] = torch.cat([raw['throttle_brake'], raw['steering']], -1)
    # the model is in train() mode, so RSSM.forward draws its "feed the prior sample forward" coin (transition.py:118-124): fed
    # explicitly, all above the dropout probability (posterior everywhere), as the oracle / product are told with use_prior
    with torch.no_grad(), G.NoisePatch(torch.zeros(b, s, 2, 1), [1.0] * s):
        batch = trainer.preprocess({k: v.clone() for k, v in raw.items()})
        out, state_dict = model(batch, deployment=True)
    fx = dict(b=b, s=s, seed=seed, outputs={})
    samples = {}
    flat = {k: out[k] for k in KEYS}
    for grp in ('prior', 'posterior'):
        for k, v in state_dict[grp].items():
            flat[f'{grp}.{k}'] = v
            assert v.shape[1] == 1, (grp, k, v.shape)
    for k, v in flat.items():
        st, smp = G.tensor_stats(v)
        st['shape'] = list(v.shape)
        fx['outputs'][k] = st
        samples['out.' + k] = smp
    from oracle import muvo_ref as R
    om = R.MileRef()
    om.load_state_dict(state0, strict=True)
    om.train()
    om.set_dropout(0.0)
    with torch.no_grad():
        pb = R.preprocess({k: v.clone() for k, v in raw.items()}, om.cfg)
        pb['action'] = raw['action']
        o = om.forward_deployment(pb)
    oflat = {k: o[k] for k in KEYS}
    for grp in ('prior', 'posterior'):
        for k, v in o[grp].items():
            oflat[f'{grp}.{k}'] = v
    for k in flat:
        print(f'  {k}: {float((oflat[k] - flat[k]).abs().max()):.3e}')
    dev = max(float((oflat[k] - flat[k]).abs().max()) for k in flat)
    print('oracle vs reference forward(deployment=True): max abs deviation', dev)
    fx['oracle_vs_reference'] = dev
    with open(os.path.join(REPO, 'tests', 'golden', 'deploy_fwd_b2s3.json'), 'w') as f:
        json.dump(fx, f)
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'deploy_fwd_b2s3_samples.npz'), **samples)
    print('wrote tests/golden/deploy_fwd_b2s3.*')


if __name__ == '__main__':
    main()
