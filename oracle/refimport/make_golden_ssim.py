"""Golden fixture of the SSIM training loss (LOSSES.SSIM=True: trainer.py:312-318 with SSIMLoss,
losses.py:292-348): one training step of the REAL reference at b=1, s=2; checks the oracle restatement and writes
tests/golden/ssim_b1s2.json.

Usage: python oracle/refimport/make_golden_ssim.py
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as G  # noqa: E402
from muvo_amd.data.synthetic import make_batch, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402


def main():
    b, s, seed = 1, 2, 8642
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = G.import_reference()
    cfg = G.effective_cfg_dict(ref_config)
    cfg['RECEPTIVE_FIELD'], cfg['FUTURE_HORIZON'], cfg['STEPS'] = s, 0, 100000
    cfg['LOSSES']['SSIM'] = True
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
    print(f'reference step with the SSIM loss {time.time() - t0:.1f}s total={total.item():.6f}; {len(losses)} losses')
    fx = dict(b=b, s=s, seed=seed, use_prior=use_prior, total=float(total), losses={k: float(v) for k, v in losses.items()},
              grad_l2={n: float(p.grad.double().pow(2).sum().sqrt()) for n, p in model.named_parameters()
                       if p.grad is not None and n.startswith('rgb_decoder.')})
    from oracle import muvo_ref
    om = muvo_ref.MileRef(cfg={**muvo_ref.base_1d_cfg(), 'SSIM': True})
    om.load_state_dict(model.state_dict(), strict=True)
    om.train()
    om.set_dropout(0.0)
    o_total, o_losses, _, _ = muvo_ref.training_step(om, raw, eps, use_prior)
    o_total.backward()
    assert set(o_losses) == set(fx['losses'])
    dev = max(abs(float(o_losses[k]) - fx['losses'][k]) / max(abs(fx['losses'][k]), 1e-12) for k in fx['losses'])
    gdev = max(abs(float(p.grad.double().pow(2).sum().sqrt()) - fx['grad_l2'][n]) / fx['grad_l2'][n]
               for n, p in om.named_parameters() if n in fx['grad_l2'] and fx['grad_l2'][n] > 0)
    print(f'oracle vs reference: max rel loss dev {dev:.3e}, max rel grad-norm dev {gdev:.3e}')
    fx['oracle_vs_reference'] = dict(max_rel_loss_dev=dev, max_rel_grad_norm_dev=gdev)
    with open(os.path.join(REPO, 'tests', 'golden', 'ssim_b1s2.json'), 'w') as f:
        json.dump(fx, f)
    print('wrote tests/golden/ssim_b1s2.json')


if __name__ == '__main__':
    main()
