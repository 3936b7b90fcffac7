"""Data-parallel golden fixture (runs ONLY in the build container, where /root/reference exists).

What two data-parallel ranks of the REAL reference compute (Lightning DDP semantics, train.py:93-98: every rank runs
forward/backward on its own batch with its own BatchNorm statistics, gradients are averaged over ranks, every rank takes
the same AdamW step): the reference model is run on rank 0's batch and on rank 1's batch from the same weights, the two
gradient sets are averaged, `optimizer.step()` is applied.  Written to tests/golden/base1d_dp2_b1s2.json: the per-rank
losses, the L2 norm of every averaged gradient and the parameter checksums after each of the steps.  The HIP path must
reproduce them with flat-gradient accumulation + grad_scale = 1/2 (tests/test_dp_gpu.py, one GPU), which is the arithmetic
of `SegmentedGradReducer` (sum all-reduce) + `FusedAdamW.grad_scale`.

Usage: python oracle/refimport/make_golden_dp.py [--world 2 --b 1 --s 2 --steps 2]
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REPO, NoisePatch, effective_cfg_dict, import_reference  # noqa: E402

from muvo_amd.data.synthetic import make_batch, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402


def rank_seed(seed, rank, step):
    return seed + 10 * rank + step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=2)
    ap.add_argument('--b', type=int, default=1)
    ap.add_argument('--s', type=int, default=2)
    ap.add_argument('--seed', type=int, default=1234)
    ap.add_argument('--steps', type=int, default=2)
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = import_reference()
    cfg_dict = effective_cfg_dict(ref_config)
    cfg_dict['RECEPTIVE_FIELD'], cfg_dict['FUTURE_HORIZON'], cfg_dict['STEPS'] = args.s, 0, 100000
    trainer = ref_trainer.WorldModelTrainer(cfg_dict)
    trainer.train()
    trainer.preprocess.eval()
    model = trainer.model
    detinit.fill_state_dict_(model)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    opts, scheds = trainer.configure_optimizers()
    opt, sched = opts[0], scheds[0]['scheduler']
    fixture = dict(world=args.world, b=args.b, s=args.s, seed=args.seed, steps=[],
                   batch_seed='seed + 10*rank + step', noise_seed='seed + 10*rank')
    params = dict(model.named_parameters())
    for step in range(args.steps):
        rec = dict(ranks=[])
        acc = {}
        for rank in range(args.world):
            eps, use_prior = make_noise(args.b, args.s, seed=rank_seed(args.seed, rank, 0))
            coin = detinit.uniform_01(detinit.name_key(f'noise:{rank_seed(args.seed, rank, 0)}') + 7, args.s)
            batch = make_batch(args.b, args.s, seed=rank_seed(args.seed, rank, step))
            t0 = time.time()
            with NoisePatch(eps, coin):
                output, _ = trainer.forward(batch)
            losses = trainer.compute_loss(batch, output)
            total = trainer.loss_reducing(losses)
            opt.zero_grad(set_to_none=True)
            total.backward()
            print(f'step {step} rank {rank}: fwd+bwd {time.time() - t0:.1f}s total={total.item():.6f}')
            rec['ranks'].append(dict(total=float(total.item()), losses={k: float(v.item()) for k, v in losses.items()},
                                     use_prior=use_prior))
            for n, p in params.items():
                if p.grad is not None:
                    acc[n] = p.grad.detach().clone() if n not in acc else acc[n] + p.grad
        for n, p in params.items():
            p.grad = (acc[n] / args.world) if n in acc else None
        rec['avg_grad_l2'] = {n: (float(p.grad.double().pow(2).sum().sqrt()) if p.grad is not None else None)
                              for n, p in params.items()}
        rec['lr'] = [g['lr'] for g in opt.param_groups]
        opt.step()
        sched.step()
        rec['param_checksums_after_step'] = {n: [float(p.detach().double().sum()), float(p.detach().double().abs().sum())]
                                             for n, p in params.items()}
        fixture['steps'].append(rec)
    with open(os.path.join(REPO, 'tests', 'golden', f'base1d_dp{args.world}_b{args.b}s{args.s}.json'), 'w') as f:
        json.dump(fixture, f)
    print('wrote DP fixture')


if __name__ == '__main__':
    main()
