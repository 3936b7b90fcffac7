"""Rounding sensitivity of the REAL reference on the base_1d golden step (same weights, batch and RSSM noise as step 0 of
tests/golden/base1d_b1s2.json): the step is run twice, plain and with every convolution output multiplied by
(1 + 4e-6 * N(0,1)) - the size of the bf16x3 split-product error of the HIP convolution kernels.  ReLU, max-pool and
L1-sign decisions that flip under such a perturbation move the gradients by far more than the perturbation itself; the
per-parameter L2 distance between the two gradients is the floor below which a gradient comparison says nothing about the
implementation.  Writes tests/golden/base1d_b1s2_rounding.json (data only).

Usage: python oracle/refimport/make_rounding_sensitivity.py [b1s2|b2s4]
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as G  # noqa: E402
from muvo_amd.data.synthetic import make_batch, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402

REL = 4e-6


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'b1s2'
    fx = json.load(open(os.path.join(REPO, 'tests', 'golden', f'base1d_{tag}.json')))
    b, s, seed = fx['b'], fx['s'], fx['seed']
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = G.import_reference()
    cfg = G.effective_cfg_dict(ref_config)
    cfg['RECEPTIVE_FIELD'], cfg['FUTURE_HORIZON'], cfg['STEPS'] = s, 0, 100000
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
    eps, _ = make_noise(b, s, seed=seed)
    coin = detinit.uniform_01(detinit.name_key(f'noise:{seed}') + 7, s)

    def step():
        for p in model.parameters():
            p.grad = None
        batch = make_batch(b, s, seed=seed)
        with G.NoisePatch(eps, coin):
            output, _ = trainer.forward(batch)
        losses = trainer.compute_loss(batch, output)
        trainer.loss_reducing(losses).backward()
        return ({k: float(v) for k, v in losses.items()},
                {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None})

    l0, g0 = step()
    ref = fx['steps'][0]
    assert max(abs(l0[k] - v) / max(abs(v), 1e-12) for k, v in ref['losses'].items()) < 1e-6, 'not the fixture step'
    gen = torch.Generator().manual_seed(99)
    hooks = [m.register_forward_hook(lambda mod, inp, out: out * (1.0 + REL * torch.randn(out.shape, generator=gen)))
             for m in model.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.ConvTranspose2d))]
    l1, g1 = step()
    for h in hooks:
        h.remove()
    out = dict(rel_perturbation=REL, n_perturbed_layers=len(hooks),
               losses={k: abs(l1[k] - l0[k]) / max(abs(l0[k]), 1e-12) for k in l0},
               grad_l2_err={n: float((g1[n] - g0[n]).pow(2).sum().sqrt()) for n in g0},
               grad_max_err={n: float((g1[n] - g0[n]).abs().max()) for n in g0})
    rel = {n: out['grad_l2_err'][n] / max(float(g0[n].pow(2).sum().sqrt()), 1e-30) for n in g0}
    print(f'{len(hooks)} conv layers perturbed by {REL}: max rel loss change {max(out["losses"].values()):.2e}; '
          f'relative gradient change: median {sorted(rel.values())[len(rel) // 2]:.2e}, max {max(rel.values()):.2e}')
    for n, v in sorted(rel.items(), key=lambda kv: -kv[1])[:10]:
        print(f'   {n:60s} {v:.2e}')
    with open(os.path.join(REPO, 'tests', 'golden', f'base1d_{tag}_rounding.json'), 'w') as f:
        json.dump(out, f)
    print(f'wrote tests/golden/base1d_{tag}_rounding.json')


if __name__ == '__main__':
    main()
