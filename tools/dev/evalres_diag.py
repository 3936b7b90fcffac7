"""dev: gradient-norm deviations of tests/test_eval_resolution.py::test_hip_eval_resolution_matches_reference, worst ten tensors"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import test_eval_resolution as T
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch, make_noise
from muvo_amd.trainer import WorldModelTrainer
from muvo_amd.utils import detinit
from muvo_amd import ops
if len(sys.argv) > 1 and sys.argv[1] == 'f32':
    ops.set_conv_mode(ops.CONV_F32)
dev = torch.device('cuda:0')
fx, smp = T._fixture()
b, s = fx['b'], fx['s']
cfg = base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000)
cfg.EVAL.RESOLUTION.ENABLED, cfg.EVAL.RESOLUTION.FACTOR, cfg.EVAL.RGB_SUPERVISION = True, 2, False
tr = WorldModelTrainer(cfg.convert_to_dict(), device=dev)
tr.train(); tr.preprocess.augment = False
detinit.fill_state_dict_(tr.model)
for layer in tr.model.transformer_encoder.layers:
    layer.p = 0.0
opts, _ = tr.configure_optimizers()
eps, use_prior = make_noise(b, s, seed=fx['seed'])
batch = make_batch(b, s, seed=fx['seed'], device=dev)
opts[0].zero_grad()
losses, output, _, _ = tr.shared_step(batch, mode='train', noise=eps.to(dev), use_prior=use_prior)
tr.loss_reducing(losses).backward()
named = dict(tr.model.named_parameters())
dev_ = sorted(((abs(named[n].grad.double().pow(2).sum().sqrt().item() - ref) / max(ref, 1e-12), n, ref) for n, ref in fx['grad_l2'].items()), reverse=True)
print('loss dev max %.2e' % max(abs(losses[k].item() - v) / max(abs(v), 1e-12) for k, v in fx['losses'].items()))
for d, n, ref in dev_[:8]:
    print('%.2e  %s  (ref %.3e)' % (d, n, ref))
