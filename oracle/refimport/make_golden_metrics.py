"""Golden fixture of the evaluation metrics: feeds deterministic synthetic batches through
the REAL reference's metric classes (muvo/metrics.py:47-317 via muvo/trainer.py:426-490) in the build container through
the import stubs and writes tests/golden/metrics.json (expected statistics only; inputs are regenerated from
muvo_amd/data/metric_inputs.py).

Usage: python oracle/refimport/make_golden_metrics.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import make_golden as G  # noqa: E402
from muvo_amd.data.metric_inputs import metric_case  # noqa: E402

SCALE = 50.0  # LIDAR_RE.SCALE (muvo.yml:67-68)


def main():
    G.import_reference()
    import muvo.metrics as M
    ssim, psnr, cd, ssc = M.SSIMMetric(channel=3), M.PSNRMetric(max_pixel_val=1.0), M.CDMetric(), M.SSCMetrics(2)
    per_batch = []
    for k in range(3):
        c = metric_case(k)
        ssim.add_batch(prediction=c['rgb_pred'], target=c['rgb_target'])
        psnr.add_batch(prediction=c['rgb_pred'], target=c['rgb_target'])
        # trainer.py:450-456
        pcd_t = c['rv_target'].permute(0, 1, 3, 4, 2).flatten(2, 3).flatten(0, 1) * SCALE
        pcd_p = c['rv_pred'].permute(0, 1, 3, 4, 2).flatten(2, 3).flatten(0, 1) * SCALE
        cd.add_batch(pcd_p[:, c['cd_index'], :-1], pcd_t[:, c['cd_index'], :-1])
        # trainer.py:482-490
        b, s, cc, x, y, z = c['voxel_logits'].shape
        y_pred = torch.argmax(c['voxel_logits'].reshape(b * s, cc, x, y, z), dim=1)
        ssc.add_batch(y_pred, c['voxel_label'].reshape(b * s, x, y, z))
        st = ssc.get_stats()
        per_batch.append(dict(ssim=float(ssim.get_stat()), psnr=float(psnr.get_stat()), cd=float(cd.get_stat()),
                              ssc=dict(precision=float(st['precision']), recall=float(st['recall']), iou=float(st['iou']),
                                       iou_ssc=[float(v) for v in st['iou_ssc']], iou_ssc_mean=float(st['iou_ssc_mean']),
                                       completion=[int(ssc.completion_tp), int(ssc.completion_fp), int(ssc.completion_fn)],
                                       tps=[int(v) for v in ssc.tps], fps=[int(v) for v in ssc.fps],
                                       fns=[int(v) for v in ssc.fns])))
        print(k, per_batch[-1])
    with open(os.path.join(REPO, 'tests', 'golden', 'metrics.json'), 'w') as f:
        json.dump(dict(scale=SCALE, n_classes=2, after_batch=per_batch), f, indent=1)
    print('wrote tests/golden/metrics.json')


if __name__ == '__main__':
    main()
