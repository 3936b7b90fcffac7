"""Fixture of EVAL.MASK_VIEW (runs ONLY in the build container): the bird's-eye-view out-of-view mask of the REAL reference
(muvo/utils/geometry_utils.py:37-61 `get_out_of_view_mask`, applied by PreProcess.prepare_bev_labels, preprocess.py:20-21,52-54,
70-72) for the default configuration and two variations (field of view, forward offset).  The reference function uses the
`np.bool` alias that numpy >= 1.24 removed (its requirements pin an older numpy): the alias is restored here before the call.
Writes tests/golden/maskview.npz.   Usage: python oracle/refimport/make_golden_maskview.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402


def main():
    if not hasattr(np, 'bool'):
        np.bool = bool
    ref_trainer, ref_config = G.import_reference()
    from muvo.utils.geometry_utils import get_out_of_view_mask
    out = {}
    for tag, over in (('default', {}), ('fov60', {'IMAGE.FOV': 60}), ('offset', {'BEV.OFFSET_FORWARD': -32, 'IMAGE.FOV': 90})):
        cfg = ref_config.get_cfg()
        cfg.defrost() if hasattr(cfg, 'defrost') else None
        for k, v in over.items():
            node = cfg
            parts = k.split('.')
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v
        m = np.asarray(get_out_of_view_mask(cfg)).astype(bool)
        out[f'{tag}_bits'] = np.packbits(m)
        out[f'{tag}_shape'] = np.array(m.shape)
        out[f'{tag}_cfg'] = np.array([cfg.IMAGE.FOV, cfg.IMAGE.SIZE[1], cfg.BEV.RESOLUTION, cfg.IMAGE.CROP[0], cfg.IMAGE.CROP[2], cfg.BEV.SIZE[0],
                                      cfg.BEV.SIZE[1], cfg.BEV.OFFSET_FORWARD, cfg.IMAGE.CAMERA_POSITION[0]], dtype=np.float64)
        print(tag, m.shape, int(m.sum()), 'masked cells')
    np.savez_compressed(os.path.join(G.REPO, 'tests', 'golden', 'maskview.npz'), **out)
    print('wrote tests/golden/maskview.npz')


if __name__ == '__main__':
    main()
