"""Golden fixture of the loss options that are off in base_1d (leftovers): one training step of the
REAL reference with VOXEL_SEG.{N_CLASSES=9, USE_WEIGHTS, USE_TOP_K} (VoxelLoss with constants.VOXEL_SEG_WEIGHTS and the top-k
selection, muvo/losses.py:144-186) and LOSSES.RGB_INSTANCE (the instance-masked second RGB term, muvo/trainer.py:303-321,
muvo/models/preprocess.py:115-125) at b=1, s=2; checks the oracle restatement and writes tests/golden/lossopts_b1s2.{json,npz}.

Usage: python oracle/refimport/make_golden_lossopts.py
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
from muvo_amd.data.synthetic import make_batch, make_image_instance_mask, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402


def main():
    b, s, seed = 1, 2, 2468
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = G.import_reference()
    cfg = G.effective_cfg_dict(ref_config)
    cfg['RECEPTIVE_FIELD'], cfg['FUTURE_HORIZON'], cfg['STEPS'] = s, 0, 100000
    cfg['VOXEL_SEG'].update(N_CLASSES=9, USE_WEIGHTS=True, USE_TOP_K=True, TOP_K_RATIO=0.25)
    cfg['LOSSES']['RGB_INSTANCE'] = True
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
    batch = make_batch(b, s, seed=seed, n_voxel_classes=9)
    batch['image_instance_mask'] = make_image_instance_mask(b, s, seed)
    raw = {k: v.clone() for k, v in batch.items()}
    t0 = time.time()
    with G.NoisePatch(eps, coin):
        output, _ = trainer.forward(batch)
    losses = trainer.compute_loss(batch, output)
    total = trainer.loss_reducing(losses)
    total.backward()
    print(f'reference step {time.time() - t0:.1f}s total={total.item():.6f}; {len(losses)} losses')
    fx = dict(b=b, s=s, seed=seed, use_prior=use_prior, total=float(total), losses={k: float(v) for k, v in losses.items()},
              cfg=dict(VOXEL_SEG=cfg['VOXEL_SEG'], RGB_INSTANCE=True), outputs={}, grad_l2={})
    samples = {}
    for k in ['voxel_1', 'voxel_4', 'rgb_1']:
        st, smp = G.tensor_stats(output[k])
        st['shape'] = list(output[k].shape)
        fx['outputs'][k] = st
        samples['out.' + k] = smp
    for k in ['image_instance_mask_1', 'image_instance_mask_2', 'image_instance_mask_4', 'voxel_label_4']:
        st, smp = G.tensor_stats(batch[k].float())
        st['shape'] = list(batch[k].shape)
        fx['outputs']['batch.' + k] = st
        samples['batch.' + k] = smp
    for n, p in model.named_parameters():
        if p.grad is not None and (n.startswith(('voxel_decoder.', 'rgb_decoder.')) or n in ('rssm.recurrent_model.weight_hh',)):
            fx['grad_l2'][n] = float(p.grad.double().pow(2).sum().sqrt())
    from oracle import muvo_ref
    ocfg = dict(muvo_ref.base_1d_cfg(), VOXEL_N_CLASSES=9, VOXEL_USE_WEIGHTS=True, VOXEL_USE_TOP_K=True, VOXEL_TOP_K_RATIO=0.25,
                RGB_INSTANCE=True)
    om = muvo_ref.MileRef(ocfg)
    om.load_state_dict(model.state_dict(), strict=True)
    om.train()
    om.set_dropout(0.0)
    o_total, o_losses, o_out, _ = muvo_ref.training_step(om, raw, eps, use_prior)
    o_total.backward()
    assert set(o_losses) == set(fx['losses']), set(o_losses) ^ set(fx['losses'])
    dev = max(abs(float(o_losses[k]) - fx['losses'][k]) / max(abs(fx['losses'][k]), 1e-12) for k in fx['losses'])
    gdev = max(abs(float(p.grad.double().pow(2).sum().sqrt()) - fx['grad_l2'][n]) / fx['grad_l2'][n]
               for n, p in om.named_parameters() if n in fx['grad_l2'] and fx['grad_l2'][n] > 0)
    print(f'oracle vs reference: max rel loss dev {dev:.3e}, max rel grad-norm dev {gdev:.3e}')
    fx['oracle_vs_reference'] = dict(max_rel_loss_dev=dev, max_rel_grad_norm_dev=gdev)
    with open(os.path.join(REPO, 'tests', 'golden', 'lossopts_b1s2.json'), 'w') as f:
        json.dump(fx, f, default=list)
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'lossopts_b1s2_samples.npz'), **samples)
    print('wrote tests/golden/lossopts_b1s2.*')


if __name__ == '__main__':
    main()
