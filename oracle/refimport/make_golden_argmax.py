"""Adds the FULL voxel argmax of the reference and its decision-margin classes to an existing base1d fixture (runs ONLY in the
build container).  One forward pass of the REAL reference on step 0 of tests/golden/base1d_<tag>.json; checks that it
reproduces the fixture's losses and argmax digest bit for bit, then writes tests/golden/base1d_<tag>_argmax.npz:
  argmax_bits        packed argmax of voxel_1 (class 1 = bit set), every voxel
  margin_lt_{2e-3,1e-2,5e-2}_bits   packed masks of the voxels whose top-2 logit margin is below that value
so that the GPU tests can say exactly how many voxels decide differently and how decisive the reference was about them.
Usage: python oracle/refimport/make_golden_argmax.py [b1s2|b2s4]"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from muvo_amd.data.synthetic import make_batch, make_noise  # noqa: E402
from muvo_amd.utils import detinit  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'b1s2'
    fx = json.load(open(os.path.join(G.REPO, 'tests', 'golden', f'base1d_{tag}.json')))
    b, s, seed = fx['b'], fx['s'], fx['seed']
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_trainer, ref_config = G.import_reference()
    cfg = G.effective_cfg_dict(ref_config)
    cfg['RECEPTIVE_FIELD'], cfg['FUTURE_HORIZON'], cfg['STEPS'] = s, 0, 100000
    trainer = ref_trainer.WorldModelTrainer(cfg)
    trainer.train()
    trainer.preprocess.eval()
    detinit.fill_state_dict_(trainer.model)
    for m in trainer.model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    eps, _ = make_noise(b, s, seed=seed)
    coin = detinit.uniform_01(detinit.name_key(f'noise:{seed}') + 7, s)
    batch = make_batch(b, s, seed=seed)
    with torch.no_grad(), G.NoisePatch(eps, coin):
        output, _ = trainer.forward(batch)
        losses = trainer.compute_loss(batch, output)
    g = fx['steps'][0]
    assert all(float(losses[k]) == v for k, v in g['losses'].items()), 'not the fixture step'
    v1 = output['voxel_1']
    am = v1.argmax(dim=2).reshape(-1).to(torch.uint8).numpy()
    assert hashlib.sha256(np.packbits(am.astype(bool)).tobytes()).hexdigest() == g['voxel_1_argmax_sha256']
    top2 = v1.topk(2, dim=2).values
    margin = (top2[:, :, 0] - top2[:, :, 1]).reshape(-1).numpy()
    out = {'argmax_bits': np.packbits(am.astype(bool))}
    for name, t in (('2e-3', 2e-3), ('1e-2', 1e-2), ('5e-2', 5e-2)):
        out[f'margin_lt_{name}_bits'] = np.packbits(margin < t)
        print(f'margin < {name}: {int((margin < t).sum())} of {margin.size} voxels')
    out['logit_absmax'] = np.float32(v1.abs().max())
    np.savez_compressed(os.path.join(G.REPO, 'tests', 'golden', f'base1d_{tag}_argmax.npz'), **out)
    print('wrote', f'tests/golden/base1d_{tag}_argmax.npz', 'logit |max|', float(v1.abs().max()))


if __name__ == '__main__':
    main()
