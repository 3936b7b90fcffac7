"""Golden fixture of the bird's-eye-view segmentation head (SEMANTIC_SEG=True: BevDecoder +
SegmentationHead common.py:147-224,249-271,370-424, label preparation preprocess.py:50-100 with
convert_instance_mask_to_center_and_offset_label instance_utils.py:4-35, losses trainer.py:266-291): one training step of
the REAL reference at b=1, s=2; checks the oracle restatement and writes tests/golden/bevseg_b1s2.{json,npz}.

Usage: python oracle/refimport/make_golden_bevseg.py
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
from muvo_amd.data.synthetic import make_batch, make_bev_labels, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402


def main():
    b, s, seed = 1, 2, 97531
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = G.import_reference()
    cfg = G.effective_cfg_dict(ref_config)
    cfg['RECEPTIVE_FIELD'], cfg['FUTURE_HORIZON'], cfg['STEPS'] = s, 0, 100000
    cfg['SEMANTIC_SEG']['ENABLED'] = True
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
    batch.update(make_bev_labels(b, s, seed))
    raw = {k: v.clone() for k, v in batch.items()}
    t0 = time.time()
    with G.NoisePatch(eps, coin):
        output, _ = trainer.forward(batch)
    losses = trainer.compute_loss(batch, output)
    total = trainer.loss_reducing(losses)
    total.backward()
    print(f'reference step with the BEV head {time.time() - t0:.1f}s total={total.item():.6f}; {len(losses)} losses')
    fx = dict(b=b, s=s, seed=seed, use_prior=use_prior, total=float(total), losses={k: float(v) for k, v in losses.items()},
              state_dict={k: list(v.shape) for k, v in model.state_dict().items()}, outputs={},
              grad_l2={n: float(p.grad.double().pow(2).sum().sqrt()) for n, p in model.named_parameters()
                       if p.grad is not None and n.startswith('bev_decoder.')},
              cfg=dict(SEMANTIC_SEG=cfg['SEMANTIC_SEG'], INSTANCE_SEG=cfg['INSTANCE_SEG']))
    samples = {}
    for k in ['bev_segmentation_1', 'bev_segmentation_4', 'bev_instance_center_1', 'bev_instance_offset_2']:
        st, smp = G.tensor_stats(output[k])
        st['shape'] = list(output[k].shape)
        fx['outputs'][k] = st
        samples['out.' + k] = smp
    for k in ['birdview_label_4', 'center_label_1', 'center_label_4', 'offset_label_1', 'offset_label_2']:
        st, smp = G.tensor_stats(batch[k].float())
        st['shape'] = list(batch[k].shape)
        fx['outputs']['batch.' + k] = st
        samples['batch.' + k] = smp
    from oracle import muvo_ref
    om = muvo_ref.MileRef(aux_heads=('bev',))
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
    with open(os.path.join(REPO, 'tests', 'golden', 'bevseg_b1s2.json'), 'w') as f:
        json.dump(fx, f, default=list)
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'bevseg_b1s2_samples.npz'), **samples)
    print('wrote tests/golden/bevseg_b1s2.*')


if __name__ == '__main__':
    main()
