"""Evaluation metrics of the validation path on the GPU: the reference's SSIMMetric, PSNRMetric,
CDMetric and SSCMetrics (muvo/metrics.py:47-317) as driven by WorldModelTrainer.add_metrics / compute_ssc_metrics
(muvo/trainer.py:426-490), computed by the HIP kernels of csrc/metrics.hip through the C ABI.  Same class names, methods
and running-average semantics (including the reference's `count = 1e-8` start value); no CPU fallback."""
import ctypes as C

import numpy as np
import torch

from . import ops


def _gauss_window(window_size=11, sigma=1.5):
    """losses.py:304-314: the 2-D window exactly as the reference builds it (fp32 outer product of the normalised 1-D
    Gaussian)."""
    x = torch.arange(window_size)
    g = torch.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2))
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float().contiguous()


def ssim_frames(prediction, target, window_size=11, sigma=1.5, L=1.0):
    """SSIMLoss._ssim (losses.py:316-339): per-frame mean of the SSIM map.  (b, s, c, h, w) -> (b*s,) float32."""
    assert window_size == 11, 'the kernel is built for the reference window (11)'
    b, s, c, h, w = prediction.shape
    p, t = prediction.float().contiguous(), target.float().contiguous()
    win = _gauss_window(window_size, sigma).to(p.device)
    sums = torch.zeros(b * s, dtype=torch.float64, device=p.device)
    ops._ck(ops.lib().muvo_ssim_frames(ops._f(p), ops._f(t), ops._f(win), ops._p(sums), b * s, c, h, w,
                                       ops._fl((0.01 * L) ** 2), ops._fl((0.03 * L) ** 2), ops._st()))
    return (sums / (c * (h - window_size + 1) * (w - window_size + 1))).float()


def psnr_frames(prediction, target, max_pixel_val=1.0):
    """PSNRMetric.psnr (metrics.py:305-309): (b, s, c, h, w) -> (b, s)."""
    b, s = prediction.shape[:2]
    L = prediction[0, 0].numel()
    p, t = prediction.float().contiguous(), target.float().contiguous()
    sums = torch.zeros(b * s, dtype=torch.float64, device=p.device)
    ops._ck(ops.lib().muvo_sqdiff_frames(ops._f(p), ops._f(t), ops._p(sums), b * s, ops._i64(L), ops._st()))
    mse = (sums / L).float().view(b, s)
    return 20 * torch.log10(max_pixel_val / torch.sqrt(mse))


def chamfer_frames(prediction, target):
    """CDMetric.add_batch with reducer = mean (metrics.py:243-249): (n, P, 3), (n, Q, 3) -> (n,)."""
    n, P, _ = prediction.shape
    Q = target.shape[1]
    a, b = prediction.float().contiguous(), target.float().contiguous()
    sums = torch.zeros(n, 2, dtype=torch.float64, device=a.device)
    ops._ck(ops.lib().muvo_chamfer_sums(ops._f(a), ops._f(b), ops._p(sums), n, P, Q, ops._st()))
    # dist.min(1) runs over the prediction points (one value per target point), dist.min(2) over the target points
    return ((sums[:, 1] / Q + sums[:, 0] / P) / 2).float()


def ssc_counts(logits, label, n_classes):
    """argmax + SSCMetrics counts (trainer.py:482-490, metrics.py:77-100,143-214).  logits (n, C, x, y, z) float32, label
    (n, x, y, z) uint8 with 255 = ignore -> (completion[3], tps[C], fps[C], fns[C]) int64 device tensors."""
    n, c = logits.shape[:2]
    assert c == n_classes
    V = logits[0, 0].numel()
    lg, lb = logits.float().contiguous(), label.to(torch.uint8).contiguous()
    counts = torch.zeros(3 + 3 * c, dtype=torch.int64, device=lg.device)
    ops._ck(ops.lib().muvo_ssc_counts(ops._f(lg), ops._p(lb), ops._p(counts), ops._i64(n), c, ops._i64(V), ops._st()))
    per = counts[3:].view(c, 3)
    return counts[:3], per[:, 0], per[:, 1], per[:, 2]


class SSIMMetric:
    """metrics.py:219-235."""

    def __init__(self, channel=3, window_size=11, sigma=1.5, L=1, non_negative=False):
        self.window_size, self.sigma, self.L, self.non_negative = window_size, sigma, L, non_negative
        self.reset()

    def add_batch(self, prediction, target):
        self.count += 1
        v = ssim_frames(prediction, target, self.window_size, self.sigma, self.L)
        if self.non_negative:
            v = torch.relu(v)
        self.ssim_score += v.mean()
        self.ssim_avg = self.ssim_score / self.count

    def get_stat(self):
        return self.ssim_avg

    def reset(self):
        self.ssim_score, self.count, self.ssim_avg = 0, 1e-8, 0


class PSNRMetric:
    """metrics.py:295-317."""

    def __init__(self, max_pixel_val=1.0):
        self.max_pixel_value = max_pixel_val
        self.reset()

    def add_batch(self, prediction, target):
        self.count += 1
        self.total_psnr += self.psnr(prediction, target).mean()
        self.avg_psnr = self.total_psnr / self.count

    def psnr(self, prediction, target):
        return psnr_frames(prediction, target, self.max_pixel_value)

    def get_stat(self):
        return self.avg_psnr

    def reset(self):
        self.total_psnr, self.count, self.avg_psnr = 0, 1e-8, 0


class CDMetric:
    """metrics.py:238-258 (reducer fixed to the mean the trainer uses)."""

    def __init__(self, reducer=torch.mean):
        assert reducer is torch.mean, 'only the mean reducer of the reference trainer is implemented'
        self.reset()

    def add_batch(self, prediction, target):
        self.count += 1
        self.total_cost += chamfer_frames(prediction, target).mean()
        self.avg_cost = self.total_cost / self.count

    def get_stat(self):
        return self.avg_cost

    def reset(self):
        self.total_cost, self.count, self.avg_cost = 0, 1e-8, 0


class SSCMetrics:
    """metrics.py:47-141.  add_batch takes the voxel LOGITS (the argmax of trainer.py:487 is fused into the kernel) or an
    already arg-maxed integer prediction."""

    def __init__(self, n_classes):
        self.n_classes = n_classes
        self.reset()

    def add_batch(self, y_pred, y_true, nonempty=None, nonsurface=None):
        assert nonempty is None and nonsurface is None, 'the reference trainer never passes masks'
        self.count += 1
        if not y_pred.is_floating_point():      # class indices -> one-hot "logits"
            y_pred = torch.nn.functional.one_hot(y_pred.long(), self.n_classes).movedim(-1, 1).float()
        comp, tps, fps, fns = ssc_counts(y_pred, y_true, self.n_classes)
        self._comp += comp
        self.tps += tps
        self.fps += fps
        self.fns += fns
        self.compute()

    def compute(self):
        tp, fp, fn = (int(v) for v in self._comp.tolist())
        self.completion_tp, self.completion_fp, self.completion_fn = tp, fp, fn
        if tp != 0:
            self.precision, self.recall, self.iou = tp / (tp + fp), tp / (tp + fn), tp / (tp + fp + fn)
        else:
            self.precision, self.recall, self.iou = 0, 0, 0
        self.iou_ssc = self.tps.float() / (self.tps + self.fps + self.fns + 1e-5).float()

    def get_stats(self):
        return {'precision': self.precision, 'recall': self.recall, 'iou': self.iou, 'iou_ssc': self.iou_ssc,
                'iou_ssc_mean': torch.mean(self.iou_ssc[1:])}

    def reset(self):
        dev = torch.device('cuda', torch.cuda.current_device())
        self._comp = torch.zeros(3, dtype=torch.int64, device=dev)
        self.tps, self.fps, self.fns = (torch.zeros(self.n_classes, dtype=torch.int64, device=dev) for _ in range(3))
        self.completion_tp = self.completion_fp = self.completion_fn = 0
        self.precision = self.recall = self.iou = 0
        self.count = 1e-8
        self.iou_ssc = torch.zeros(self.n_classes, dtype=torch.float32, device=dev)


class EvalMetrics:
    """The metric set of one validation dataloader on base_1d (trainer.py:426-490): ssim, psnr, cd, ssc."""

    def __init__(self, n_classes=2, scale=50.0):
        self.scale = scale
        self.ssim, self.psnr, self.cd, self.ssc = SSIMMetric(channel=3), PSNRMetric(1.0), CDMetric(), SSCMetrics(n_classes)

    def add_batch(self, rgb_pred, rgb_target, rv_pred, rv_target, cd_index, voxel_logits, voxel_label):
        self.ssim.add_batch(prediction=rgb_pred, target=rgb_target)
        self.psnr.add_batch(prediction=rgb_pred, target=rgb_target)
        pt = rv_target.permute(0, 1, 3, 4, 2).flatten(2, 3).flatten(0, 1) * self.scale      # trainer.py:450-456
        pp = rv_pred.permute(0, 1, 3, 4, 2).flatten(2, 3).flatten(0, 1) * self.scale
        idx = torch.as_tensor(cd_index, device=pp.device).long()
        self.cd.add_batch(pp[:, idx, :-1], pt[:, idx, :-1])
        b, s, c, x, y, z = voxel_logits.shape
        self.ssc.add_batch(voxel_logits.reshape(b * s, c, x, y, z), voxel_label.reshape(b * s, x, y, z))

    def stats(self):
        st = self.ssc.get_stats()
        return dict(ssim=float(self.ssim.get_stat()), psnr=float(self.psnr.get_stat()), cd=float(self.cd.get_stat()),
                    precision=st['precision'], recall=st['recall'], iou=st['iou'], iou_ssc=st['iou_ssc'],
                    iou_ssc_mean=float(st['iou_ssc_mean']),
                    completion=[self.ssc.completion_tp, self.ssc.completion_fp, self.ssc.completion_fn],
                    tps=self.ssc.tps.tolist(), fps=self.ssc.fps.tolist(), fns=self.ssc.fns.tolist())
