"""Golden fixture of EVAL.RESOLUTION (SURVEY.md section 8, config-surface leftover): one training step of the REAL reference with
EVAL.RESOLUTION.{ENABLED=True, FACTOR=2} - PreProcess.forward down-scales the cropped image with torchvision's antialiased
resize before the model sees it (muvo/models/preprocess.py:209-210,252-273) - and EVAL.RGB_SUPERVISION=False (with the RGB
decoder on, the reference's own loss compares 320 x 832 predictions with the 160 x 416 label and fails to broadcast,
muvo/trainer.py:296-302) at b=1, s=2; checks the oracle restatement and writes tests/golden/evalres_b1s2.{json,npz}.
torchvision itself is absent from the image: stubs/torchvision restates `resize(antialias=True)` as torch's
F.interpolate(mode='bilinear', antialias=True), the call torchvision 0.15.2 makes for tensors.

Usage: python oracle/refimport/make_golden_evalres.py
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as G  # noqa: E402
from muvo_amd.data.synthetic import make_batch, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402


def main():
    b, s, seed = 1, 2, 1357
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = G.import_reference()
    cfg = G.effective_cfg_dict(ref_config)
    cfg['RECEPTIVE_FIELD'], cfg['FUTURE_HORIZON'], cfg['STEPS'] = s, 0, 100000
    cfg['EVAL']['RESOLUTION'].update(ENABLED=True, FACTOR=2)
    cfg['EVAL']['RGB_SUPERVISION'] = False
    trainer = ref_trainer.WorldModelTrainer(cfg)
    trainer.train()
    trainer.preprocess.eval()
    model = trainer.model
    detinit.fill_state_dict_(model)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    eps, use_prior = make_noise(b, s, seed=seed)
    coin = detinit.uniform_01(detinit.name_key(f'noise:{seed}') + 7, s)
    batch = make_batch(b, s, seed=seed)
    raw = {k: v.clone() for k, v in batch.items()}
    t0 = time.time()
    with G.NoisePatch(eps, coin):
        output, _ = trainer.forward(batch)
    losses = trainer.compute_loss(batch, output)
    total = trainer.loss_reducing(losses)
    total.backward()
    print(f'reference step {time.time() - t0:.1f}s total={total.item():.6f}; {len(losses)} losses; image {tuple(batch["image"].shape)}')
    assert tuple(batch['image'].shape[-2:]) == (160, 416)
    fx = dict(b=b, s=s, seed=seed, use_prior=use_prior, total=float(total), losses={k: float(v) for k, v in losses.items()},
              cfg=dict(EVAL_RESOLUTION=dict(ENABLED=True, FACTOR=2), RGB_SUPERVISION=False), outputs={}, grad_l2={},
              n_parameters=sum(1 for _ in model.parameters()))
    samples = {}
    for k in ['voxel_1', 'lidar_reconstruction_1', 'throttle_brake', 'steering']:
        st, smp = G.tensor_stats(output[k])
        st['shape'] = list(output[k].shape)
        fx['outputs'][k] = st
        samples['out.' + k] = smp
    for k in ['image', 'intrinsics']:
        st, smp = G.tensor_stats(batch[k].float(), nsample=4096)
        st['shape'] = list(batch[k].shape)
        fx['outputs']['batch.' + k] = st
        samples['batch.' + k] = smp
    for n, p in model.named_parameters():
        if p.grad is not None and (n.startswith(('encoder.', 'feat_decoder.', 'image_feature_conv.')) or n in ('rssm.recurrent_model.weight_hh', 'type_embedding')):
            fx['grad_l2'][n] = float(p.grad.double().pow(2).sum().sqrt())
    from oracle import muvo_ref
    ocfg = dict(muvo_ref.base_1d_cfg(), EVAL_RESOLUTION_FACTOR=2, RGB_SUPERVISION=False)
    om = muvo_ref.MileRef(ocfg)
    om.load_state_dict(model.state_dict(), strict=True)
    om.train()
    om.set_dropout(0.0)
    o_total, o_losses, o_out, o_batch = muvo_ref.training_step(om, raw, eps, use_prior)
    o_total.backward()
    assert set(o_losses) == set(fx['losses']), set(o_losses) ^ set(fx['losses'])
    dev = max(abs(float(o_losses[k]) - fx['losses'][k]) / max(abs(fx['losses'][k]), 1e-12) for k in fx['losses'])
    gdev = max(abs(float(p.grad.double().pow(2).sum().sqrt()) - fx['grad_l2'][n]) / fx['grad_l2'][n]
               for n, p in om.named_parameters() if n in fx['grad_l2'] and fx['grad_l2'][n] > 0)
    idev = float((o_batch['image'] - batch['image']).abs().max()) if isinstance(o_batch, dict) and 'image' in o_batch else None
    print(f'oracle vs reference: max rel loss dev {dev:.3e}, max rel grad-norm dev {gdev:.3e}, max abs image dev {idev}')
    fx['oracle_vs_reference'] = dict(max_rel_loss_dev=dev, max_rel_grad_norm_dev=gdev, max_abs_image_dev=idev)
    with open(os.path.join(REPO, 'tests', 'golden', 'evalres_b1s2.json'), 'w') as f:
        json.dump(fx, f, default=list)
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'evalres_b1s2_samples.npz'), **samples)
    print('wrote tests/golden/evalres_b1s2.*')


if __name__ == '__main__':
    main()
