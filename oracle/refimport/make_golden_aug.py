"""Training-time augmentation fixture (runs ONLY in the build container, where /root/reference exists).

Runs the REAL `muvo.models.preprocess.PreProcess` in training mode (preprocess.py:201-225 with PixelAugmentation :295-333
and RouteAugmentation :336-367 active) on a seeded synthetic batch.  torchvision 0.15.2, whose tensor algorithms the
augmentation calls, is not installed here and not vendored in /root/reference: oracle/refimport/stubs/torchvision restates
them from the published sources (that part of the pin is our restatement; everything it calls — F.conv2d, F.pad(reflect),
F.grid_sample — is the real torch).  The random draws: torch.manual_seed(seed) before the reference runs; the same seed
before muvo_amd.augment.draw_*_params(), which repeats the reference's RNG call sequence and therefore reproduces its draws
(checked here: the oracle restatement fed with those tables must equal the reference output).

Written to tests/golden/augment.{json,npz}: the two parameter tables, per-frame statistics and strided samples of the
augmented image / rgb_label_1, the full augmented route maps.
Usage: python oracle/refimport/make_golden_aug.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REPO, effective_cfg_dict, import_reference  # noqa: E402

from muvo_amd import augment  # noqa: E402
from muvo_amd.data.synthetic import make_aug_batch  # noqa: E402

B, S, SEED = 6, 2, 4360
# every branch must occur in 12 frames / 6 samples: raise the route probabilities (config values, any are legal)
OVERRIDES = {'ROUTE': {'AUGMENTATION_DROPOUT': 0.15, 'AUGMENTATION_END_OF_ROUTE': 0.2, 'AUGMENTATION_SMALL_ROTATION': 0.25,
                       'AUGMENTATION_LARGE_ROTATION': 0.25}}


def frame_stats(t):
    d = t.double().flatten(2)
    return dict(sum=d.sum(-1).flatten().tolist(), l2=d.pow(2).sum(-1).sqrt().flatten().tolist())


def main():
    ref_trainer, ref_config = import_reference()
    from muvo.models.preprocess import PreProcess
    cfg_dict = effective_cfg_dict(ref_config)
    for k, v in OVERRIDES.items():
        cfg_dict[k].update(v)
    cfg = ref_config.get_cfg(cfg_dict=cfg_dict)
    pre = PreProcess(cfg)
    pre.train()
    raw = make_aug_batch(B, S, seed=SEED)
    batch = {k: v.clone() for k, v in raw.items()}
    torch.manual_seed(SEED)
    out = pre(batch)
    torch.manual_seed(SEED)
    pix = augment.draw_pixel_params(cfg, B, S)
    route = augment.draw_route_params(cfg, B, cfg.ROUTE.SIZE)
    print('pixel modes', pix[:, 0].tolist(), 'colour jitter', pix[:, 2].tolist())
    print('route modes', route[:, 0].tolist())
    assert set(pix[:, 0].tolist()) == {0.0, 1.0, 2.0} and set(route[:, 0].tolist()) == {0.0, 1.0, 2.0, 3.0}
    assert 0 < pix[:, 2].sum() < B * S

    from oracle import muvo_ref as R
    o = R.preprocess({k: v.clone() for k, v in raw.items()}, R.base_1d_cfg(), pixel_aug=pix, route_aug=route)
    dev = {}
    for k in ('image', 'route_map', 'rgb_label_1', 'rgb_label_2', 'rgb_label_4'):
        dev[k] = float((o[k] - out[k]).abs().max())
    print('oracle vs reference max abs deviation:', dev)
    assert dev['route_map'] == 0.0 and max(dev.values()) < 2e-6, dev
    assert out['rgb_label_1'].data_ptr() != out['image'].data_ptr()

    # what the un-augmented pipeline gives: which frames changed
    pre.eval()
    plain = pre({k: v.clone() for k, v in raw.items()})
    changed = (plain['rgb_label_1'] - out['rgb_label_1']).abs().flatten(2).max(-1).values.flatten()
    assert all(c == 0.0 for f, c in enumerate(changed.tolist()) if not (pix[f, 0] != 0 or pix[f, 2] != 0)), changed
    assert sum(c > 1e-3 for c in changed.tolist()) >= 8, changed      # (a blur with sigma ~0.1 changes nothing visible)
    assert float((plain['rgb_label_2'] - out['rgb_label_2']).abs().max()) == 0.0      # label pyramid is made before the augmentation

    fx = dict(b=B, s=S, seed=SEED, overrides=OVERRIDES, oracle_vs_reference=dev,
              note='batch = muvo_amd.data.synthetic.make_aug_batch(b, s, seed)',
              image=frame_stats(out['image']), rgb_label_1=frame_stats(out['rgb_label_1']), route_map=frame_stats(out['route_map']))
    stride = 97
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'augment.npz'),
                        pixel_params=pix.numpy(), route_params=route.numpy(), route_map=out['route_map'].numpy(),
                        image_sample=out['image'].flatten()[::stride].numpy(), label_sample=out['rgb_label_1'].flatten()[::stride].numpy(),
                        sample_stride=np.int64(stride))
    with open(os.path.join(REPO, 'tests', 'golden', 'augment.json'), 'w') as f:
        json.dump(fx, f)
    print('wrote tests/golden/augment.{json,npz}')


if __name__ == '__main__':
    main()
