"""Gradient-norm deviation of the BEV golden step (tests/golden/bev_b1s2.*) under the current conv policy / env:
prints the distribution of |g_hip| / |g_ref| - 1 over the fixture's parameters."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.frustum_inputs import camera_pose
from muvo_amd.data.synthetic import make_batch, make_noise
from muvo_amd.trainer import WorldModelTrainer
from muvo_amd.utils import detinit
fx = json.load(open(os.path.join(ROOT, 'tests/golden/bev_b1s2.json')))
dev = torch.device('cuda:0')
b, s = fx['b'], fx['s']
cfg = base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000)
cfg.MODEL.TRANSFORMER.BEV = True
tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev); tr.train(); tr.preprocess.augment = False
bev_intr = tr.model.frustum_pooling.bev_intrinsics.clone()
detinit.fill_state_dict_(tr.model)
tr.model.frustum_pooling.bev_intrinsics.copy_(bev_intr)
for layer in tr.model.transformer_encoder.layers:
    layer.p = 0.0
eps, use_prior = make_noise(b, s, seed=fx['seed'])
batch = make_batch(b, s, seed=fx['seed'], device=dev)
batch['extrinsics'] = camera_pose(b, s).to(dev)
losses, output, _, _ = tr.shared_step(batch, mode='train', noise=eps.to(dev), use_prior=use_prior)
tr.loss_reducing(losses).backward()
params = dict(tr.model.named_parameters())
dev_ = {n: params[n].grad.double().pow(2).sum().sqrt().item() / max(ref, 1e-30) - 1 for n, ref in fx['grad_l2'].items() if ref > 1e-9}
v = np.array(list(dev_.values()))
print(f'{len(v)} params: mean {v.mean():+.2e} median {np.median(v):+.2e} min {v.min():+.2e} max {v.max():+.2e} '
      f'frac>0 {np.mean(v > 0):.2f} frac|.|>5e-3 {np.mean(np.abs(v) > 5e-3):.2f}')
for n, d in sorted(dev_.items(), key=lambda kv: -abs(kv[1]))[:12]:
    print(f'  {n:60s} {d:+.2e}')
for k, val in fx['losses'].items():
    print(f'  loss {k:30s} {losses[k].item() / val - 1:+.2e}')
