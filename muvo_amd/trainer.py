"""WorldModelTrainer — drop-in for muvo/trainer.py's LightningModule on the MI355X-native step.

Same constructor `(hparams, path_to_conf_file=None, pretrained_path=None)`, attributes (`cfg`, `rf`, `fh`,
`preprocess`, `model`) and methods (`forward`, `compute_loss`, `shared_step(mode='train')`, `training_step`,
`configure_optimizers`, `load_pretrained_weights`) as trainer.py:26-231,251-402,511-513,1022-1073.  If
`lightning` is importable the class derives from `pl.LightningModule`, otherwise from `torch.nn.Module` with
the same hooks so the build's own loop (bench.py / train.py) can drive it.  Validation/visualisation
(trainer.py:404-1020) are outside the training hot path."""
import os

import numpy as np
import torch

from muvo_amd import ops
from muvo_amd.config import get_cfg
from muvo_amd.models.mile import Mile
from muvo_amd.models.preprocess import PreProcess
from muvo_amd.optim import FusedAdamW
from muvo_amd.param_store import ParamStore

try:  # optional: Lightning is not a dependency of the hot path
    import lightning.pytorch as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover
    pl = None
    _Base = torch.nn.Module


_DDP_MSG = ('muvo_amd.WorldModelTrainer exchanges its gradients itself (segmented RCCL all-reduce on a side stream, '
            'muvo_amd/parallel.py): run one process per GPU with torch.distributed initialised and do NOT wrap the '
            'module in DistributedDataParallel (Lightning: use a single-device strategy per process, INTEGRATION.md §5)')


def _refuse_ddp(module, args):
    tr = getattr(module, '_trainer', None) if pl is not None else None
    strategy = getattr(tr, 'strategy', None)
    if strategy is not None and isinstance(getattr(strategy, 'model', None), torch.nn.parallel.DistributedDataParallel):
        raise RuntimeError(_DDP_MSG)


class WorldModelTrainer(_Base):
    def __init__(self, hparams, path_to_conf_file=None, pretrained_path=None, device=None):
        super().__init__()
        if pl is not None:
            self.save_hyperparameters()
        self.cfg = get_cfg(cfg_dict=hparams)
        if path_to_conf_file:
            self.cfg.merge_from_file(path_to_conf_file)
        if pretrained_path:
            self.cfg.PRETRAINED.PATH = pretrained_path
        self.vis_step = -1
        self.rf = self.cfg.RECEPTIVE_FIELD
        self.fh = self.cfg.FUTURE_HORIZON
        ops.lib()  # fail loudly, before building anything, if the HIP library is missing
        self.preprocess = PreProcess(self.cfg)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is None:
            raise RuntimeError('muvo_amd needs a GPU (gfx950): no CUDA/HIP device visible and there is no CPU path')
        with torch.device(device):
            self.model = Mile(self.cfg)
        self.load_pretrained_weights()
        self.store = None
        self._optimizer = None
        self._reducer = None
        self._global_step = 0          # plain-loop counterpart of LightningModule.global_step
        self.log_fn = None             # plain loop: callable(name, value) receiving what Lightning's self.log would
        self.logged = {}
        self.accumulate_now = False    # plain loop: True while a non-final micro-batch of gradient accumulation runs
        # evaluation metrics per validation / test dataloader (trainer.py:51-55,100-129,191-197); created on first use
        # because they hold device accumulators
        self.metrics_vals = [{}, {}, {}]
        self.metrics_vals_imagine = [{}, {}, {}]
        self.metrics_tests = [{}, {}, {}]
        self.metrics_tests_imagine = [{}, {}, {}]

    # ------------------------------------------------------------------ weights
    def load_pretrained_weights(self):
        path = self.cfg.PRETRAINED.PATH
        if path:
            if os.path.isfile(path):
                checkpoint = torch.load(path, map_location='cpu')['state_dict']
                checkpoint = {key[6:]: value for key, value in checkpoint.items() if key[:5] == 'model'}
                self.model.load_state_dict(checkpoint, strict=True)
                print(f'Loaded weights from: {path}')
            else:
                raise FileExistsError(path)

    # ------------------------------------------------------------------ forward / losses
    def forward(self, batch, deployment=False, noise=None, use_prior=None):
        ops.repack_all()           # one launch refreshes every packed weight copy made stale by the optimizer step
        batch = self.preprocess(batch)
        output, state_dict = self.model.forward(batch, deployment=deployment, noise=noise, use_prior=use_prior)
        return output, state_dict

    def deployment_forward(self, batch, is_dreaming):
        """trainer.py:218-221"""
        ops.repack_all()
        batch = self.preprocess(batch)
        return self.model.deployment_forward(batch, is_dreaming)

    def shared_step(self, batch, mode='train', predict_action=False, noise=None, use_prior=None):
        """trainer.py:223-249.  mode='train': reconstruction of the whole sequence.  Otherwise: reconstruct the first
        RECEPTIVE_FIELD frames, then imagine FUTURE_HORIZON steps from the last posterior state and score them against the
        remaining frames.  noise (b, rf + N_SAMPLES*fh, 2, S) / use_prior (rf,) make the RNG explicit: [:, t < rf] feed the
        observe steps, [:, rf + k*fh + t, 0] step t of imagined sample k."""
        if mode == 'train':
            try:
                output, state_dict = self.forward(batch, noise=noise, use_prior=use_prior)
                losses = self.compute_loss(batch, output)
            except BaseException:
                ops.reset_accumulators()      # zero-between-uses buffers may hold partial sums of the interrupted step
                raise
            return losses, output, [], []
        rf, fh = self.rf, self.fh
        batch = self.preprocess(batch)
        batch_rf = {k: v[:, :rf].contiguous() for k, v in batch.items()}
        batch_fh = {k: v[:, rf:].contiguous() for k, v in batch.items()}
        output, state_dict = self.model.forward(batch_rf, deployment=False, noise=None if noise is None else noise[:, :rf],
                                                use_prior=None if use_prior is None else list(use_prior[:rf]))
        losses = self.compute_loss(batch_rf, output)
        post = state_dict['posterior']
        state_imagine = {'hidden_state': post['hidden_state'][:, -1], 'sample': post['sample'][:, -1],
                         'throttle_brake': batch_fh['throttle_brake'], 'steering': batch_fh['steering']}
        output_imagines, losses_imagines = [], []
        for k in range(self.cfg.PREDICTION.N_SAMPLES):
            # explicit noise layout: (b, rf + N_SAMPLES * fh, 2, S); sample k of the roll-out uses rows rf + k*fh ...
            nz = None if noise is None else noise[:, rf + k * fh:rf + (k + 1) * fh, 0].contiguous()
            out_i = self.model.imagine(state_imagine, predict_action=predict_action, future_horizon=fh, noise=nz)
            output_imagines.append(out_i)
            losses_imagines.append(self.compute_loss(batch_fh, out_i))
        return losses, output, losses_imagines, output_imagines

    def _metric_set(self, metrics):
        """trainer.py:100-129,191-197: the metrics base_1d enables (ssim, psnr, cd, ssc)."""
        if not metrics:
            from .metrics import CDMetric, PSNRMetric, SSCMetrics, SSIMMetric
            if self.cfg.EVAL.RGB_SUPERVISION:
                metrics['ssim'], metrics['psnr'] = SSIMMetric(channel=3), PSNRMetric(max_pixel_val=1.0)
            if self.cfg.LIDAR_RE.ENABLED:
                metrics['cd'] = CDMetric()
            if self.cfg.VOXEL_SEG.ENABLED:
                metrics['ssc'] = SSCMetrics(self.cfg.VOXEL_SEG.N_CLASSES)
        return metrics

    def add_metrics(self, metrics, batch, output, cd_index=None):
        """trainer.py:426-480 for the heads of base_1d.  cd_index: the 10000-point subset of the Chamfer metric; the
        reference draws it with np.random.randint (trainer.py:455), pass it explicitly for reproducible numbers."""
        metrics = self._metric_set(metrics)
        if self.cfg.EVAL.RGB_SUPERVISION:
            metrics['ssim'].add_batch(prediction=output['rgb_1'].detach(), target=batch['rgb_label_1'])
            metrics['psnr'].add_batch(prediction=output['rgb_1'].detach(), target=batch['rgb_label_1'])
        if self.cfg.LIDAR_RE.ENABLED:
            lidar_target = batch['range_view_label_1']
            lidar_pred = output['lidar_reconstruction_1'].detach()
            pcd_target = lidar_target.detach().permute(0, 1, 3, 4, 2).flatten(2, 3).flatten(0, 1) * self.cfg.LIDAR_RE.SCALE
            pcd_pred = lidar_pred.permute(0, 1, 3, 4, 2).flatten(2, 3).flatten(0, 1) * self.cfg.LIDAR_RE.SCALE
            if cd_index is None:
                cd_index = np.random.randint(0, pcd_target.size(-2), 10000)
            index = torch.as_tensor(cd_index, device=pcd_pred.device).long()
            metrics['cd'].add_batch(pcd_pred[:, index, :-1], pcd_target[:, index, :-1])
        if self.cfg.VOXEL_SEG.ENABLED:
            self.compute_ssc_metrics(batch, output, metrics['ssc'])

    def compute_ssc_metrics(self, batch, output, metric):
        """trainer.py:482-490; the argmax over the class logits happens inside the counting kernel."""
        y_true = batch['voxel_label_1']
        y_pred = output['voxel_1'].detach()
        b, s, c, x, y, z = y_pred.shape
        metric.add_batch(y_pred.reshape(b * s, c, x, y, z), y_true.reshape(b * s, x, y, z))

    def validation_step(self, batch, batch_idx=0, dataloader_idx=0, noise=None, use_prior=None, cd_index=None):
        """trainer.py:404-424 (visualisation/logging hooks excepted): train-mode BatchNorm, transformer nn.Dropout modules
        off (the functional attention dropout of nn.MultiheadAttention stays on), no_grad; then the
        reconstruction metrics on the observed frames and the imagination metrics on the future frames."""
        self.train()
        layers = list(self.model.transformer_encoder.layers)
        saved = [getattr(layer, 'module_dropout_off', False) for layer in layers]
        for layer in layers:
            layer.module_dropout_off = True
        try:
            with torch.no_grad():
                loss, output, loss_imagines, output_imagines = self.shared_step(batch, mode='val', predict_action=False,
                                                                                noise=noise, use_prior=use_prior)
        finally:
            for layer, v in zip(layers, saved):
                layer.module_dropout_off = v
        batch_rf = {key: value[:, :self.rf] for key, value in batch.items() if torch.is_tensor(value)}
        batch_fh = {key: value[:, self.rf:] for key, value in batch.items() if torch.is_tensor(value)}
        self.add_metrics(self.metrics_vals[dataloader_idx], batch_rf, output, cd_index)
        for output_imagine in output_imagines:
            self.add_metrics(self.metrics_vals_imagine[dataloader_idx], batch_fh, output_imagine, cd_index)
        out = {f'val{dataloader_idx}_loss': self.loss_reducing(loss),
               f'val{dataloader_idx}_loss_imagine': sum(self.loss_reducing(li) for li in loss_imagines) / len(loss_imagines)}
        return out, loss, output, loss_imagines, output_imagines

    def compute_loss(self, batch, output):
        """The reference's 21 weighted loss terms (trainer.py:251-390), computed by fused kernels."""
        cfg = self.cfg
        losses = {}
        w_act = cfg.LOSSES.WEIGHT_ACTION
        if 'throttle_brake' in output:
            losses['throttle_brake'] = ops.l1_rows_loss(output['throttle_brake'], batch['throttle_brake'], w_act, terms=True)[0]
        if 'steering' in output:
            losses['steering'] = ops.l1_rows_loss(output['steering'], batch['steering'], w_act, terms=True)[0]
        if cfg.MODEL.TRANSITION.ENABLED and 'prior' in output and 'posterior' in output:
            pr, po = output['prior'], output['posterior']
            losses['probabilistic'] = ops.kl_loss(pr['mu'], pr['sigma'], po['mu'], po['sigma'],
                                                  cfg.LOSSES.WEIGHT_PROBABILISTIC, cfg.LOSSES.KL_BALANCING_ALPHA, terms=True)[0]
        if cfg.SEMANTIC_SEG.ENABLED:              # trainer.py:266-291
            crit = self._seg_loss('bev', cfg.SEMANTIC_SEG, is_bev=True)
            ign = float(cfg.INSTANCE_SEG.IGNORE_INDEX)
            for f in (1, 2, 4):
                d = 1 / f
                losses[f'bev_segmentation_{f}'] = crit(output[f'bev_segmentation_{f}'], batch[f'birdview_label_{f}']) \
                    * (d * cfg.LOSSES.WEIGHT_SEGMENTATION)
                wc = d * cfg.LOSSES.WEIGHT_INSTANCE * cfg.INSTANCE_SEG.CENTER_LOSS_WEIGHT
                losses[f'bev_center_{f}'] = ops.spatial_losses(output[f'bev_instance_center_{f}'], batch[f'center_label_{f}'],
                                                               [(0, 1, 2, wc)])[0]
                wo = cfg.LOSSES.WEIGHT_INSTANCE * cfg.INSTANCE_SEG.OFFSET_LOSS_WEIGHT   # offsets are discounted in the labels
                losses[f'bev_offset_{f}'] = ops.spatial_losses(output[f'bev_instance_offset_{f}'], batch[f'offset_label_{f}'],
                                                               [(0, 2, 1, wo)], ign)[0]
        if cfg.EVAL.RGB_SUPERVISION:
            for f in (1, 2, 4):
                w = 0.1 * (1 / f)  # rgb_weight literal 0.1 (trainer.py:296)
                pred = output[f'rgb_{f}']
                losses[f'rgb_{f}'] = ops.spatial_losses(pred, batch[f'rgb_label_{f}'], [(0, pred.shape[2], 1, w)], terms=True)[0]
                if cfg.LOSSES.RGB_INSTANCE:       # trainer.py:303-321: + 0.5 x the same L1 over the vehicle / pedestrian pixels
                    losses[f'rgb_{f}'] = losses[f'rgb_{f}'] + ops.spatial_losses(
                        pred, batch[f'rgb_label_{f}'], [(0, pred.shape[2], 1, 0.5 * w)], mask=batch[f'image_instance_mask_{f}'])[0]
                if cfg.LOSSES.SSIM:               # trainer.py:312-318: 0.6 * (1 - mean SSIM), same rgb weight and discount
                    if '_ssim_loss' not in self.__dict__:
                        from .losses import SSIMLoss
                        self.__dict__['_ssim_loss'] = SSIMLoss(channel=3)
                    losses[f'ssim_{f}'] = (1 - self.__dict__['_ssim_loss'](pred, batch[f'rgb_label_{f}'])) * (w * 0.6)
        if cfg.LIDAR_RE.ENABLED:
            for f in (1, 2, 4):
                w = (1 / f) * cfg.LOSSES.WEIGHT_LIDAR_RE
                pred = output[f'lidar_reconstruction_{f}']
                c = pred.shape[2]
                both = ops.spatial_losses(pred, batch[f'range_view_label_{f}'], [(0, 3, 2, w), (c - 1, c, 1, w)], terms=True)
                losses[f'lidar_re_{f}'] = both[0]
                losses[f'lidar_depth_{f}'] = both[1]
        # config-off heads of base_1d (trainer.py:338-365)
        if cfg.LIDAR_SEG.ENABLED:
            crit = self._seg_loss('lidar', cfg.LIDAR_SEG)
            for f in (1, 2, 4):
                losses[f'lidar_seg_{f}'] = crit(output[f'lidar_segmentation_{f}'], batch[f'range_view_seg_label_{f}']) \
                    * ((1 / f) * cfg.LOSSES.WEIGHT_LIDAR_SEG)
        if cfg.SEMANTIC_IMAGE.ENABLED:
            crit = self._seg_loss('image', cfg.SEMANTIC_IMAGE)
            for f in (1, 2, 4):
                losses[f'semantic_image_{f}'] = crit(output[f'semantic_image_{f}'], batch[f'semantic_image_label_{f}']) \
                    * ((1 / f) * cfg.LOSSES.WEIGHT_SEM_IMAGE)
        if cfg.DEPTH.ENABLED:
            for f in (1, 2, 4):
                w = (1 / f) * cfg.LOSSES.WEIGHT_DEPTH
                losses[f'depth_{f}'] = ops.spatial_losses(output[f'depth_{f}'], batch[f'depth_label_{f}'], [(0, 1, 1, w)])[0]
        if cfg.VOXEL_SEG.ENABLED:
            for f in (1, 2, 4):
                w = (1 / f) * cfg.LOSSES.WEIGHT_VOXEL
                vs = cfg.VOXEL_SEG
                cw = None
                if vs.USE_WEIGHTS:                # constants.py:39 through VoxelLoss(use_weights) (losses.py:155-165)
                    from .losses import VOXEL_SEG_WEIGHTS
                    logits_f = output[f'voxel_{f}']
                    if len(VOXEL_SEG_WEIGHTS) != logits_f.shape[2]:      # F.cross_entropy of the reference raises here as well
                        raise RuntimeError(f'VOXEL_SEG.USE_WEIGHTS: {len(VOXEL_SEG_WEIGHTS)} class weights (constants.py:39) for '
                                           f'{logits_f.shape[2]} voxel classes')
                    cache = self.__dict__.setdefault('_voxel_class_weights', {})     # one H2D copy per device, not three per step
                    cw = cache.get(logits_f.device)
                    if cw is None:
                        cw = cache[logits_f.device] = torch.tensor(VOXEL_SEG_WEIGHTS, dtype=torch.float32, device=logits_f.device)
                # (USE_TOP_K replaces the cross-entropy term: the fused kernel then only delivers the two scaling terms)
                three = ops.voxel_losses(output[f'voxel_{f}'], batch[f'voxel_label_{f}'], w, cw, terms=True)
                if vs.USE_TOP_K:                  # losses.py:179-184: the k hardest voxels of every frame
                    from .losses import VoxelLoss
                    crit = self.__dict__.setdefault('_voxel_topk', VoxelLoss(True, vs.TOP_K_RATIO, vs.USE_WEIGHTS))
                    losses[f'voxel_{f}'] = crit(output[f'voxel_{f}'], batch[f'voxel_label_{f}']) * w
                else:
                    losses[f'voxel_{f}'] = three[0]
                losses[f'sem_scal_{f}'] = three[1]
                losses[f'geo_scal_{f}'] = three[2]
        return losses

    def _seg_loss(self, tag, c, is_bev=False):
        """trainer.py:61-66,132-161: SegmentationLoss(use_top_k, top_k_ratio, use_weights, is_bev)."""
        from .losses import SegmentationLoss
        cache = self.__dict__.setdefault('_seg_losses', {})
        if tag not in cache:
            cache[tag] = SegmentationLoss(use_top_k=c.USE_TOP_K, top_k_ratio=c.TOP_K_RATIO, use_weights=c.USE_WEIGHTS, is_bev=is_bev)
        return cache[tag]

    def loss_reducing(self, loss):
        vals = list(loss.values())
        return ops.sum_scalars(vals)

    def training_step(self, batch, batch_idx=0, noise=None, use_prior=None):
        """trainer.py:392-402: switch the RSSM to active inference at batch STEPS, shared_step('train'), log the 21 terms,
        return their sum."""
        if batch_idx == self.cfg.STEPS and self.cfg.MODEL.TRANSITION.ENABLED:
            print('!' * 50)
            print('ACTIVE INFERENCE ACTIVATED')
            print('!' * 50)
            self.model.rssm.active_inference = True
        if self._optimizer is not None and self.model.seed_epoch != self._optimizer._step + 1:
            # dropout seeds = f(rank, optimizer step, micro-batch within the step): reproducible across a checkpoint resume
            self.model.seed_epoch = self._optimizer._step + 1
            self.model._step_seed = 0
        if self._reducer is not None:
            self._reducer.accumulating = self._is_accumulating()
            self._reducer.begin_step()
        losses, output, _, _ = self.shared_step(batch, mode='train', noise=noise, use_prior=use_prior)
        # kept detached: a loss tensor with its grad_fn would keep this step's autograd graph - and the AccumulateGrad nodes of
        # every parameter, created on the side streams - alive into the next step (memory, and PyTorch's stream-mismatch warning
        # as soon as a parameter is then used on another stream; Lightning's self.log detaches as well)
        self.last_losses = {k: v.detach() for k, v in losses.items()}
        self.logging_and_visualisation(batch, output, [], losses, None, batch_idx, prefix='train')
        return self.loss_reducing(losses)

    def logging_and_visualisation(self, batch, output, output_imagine, loss, loss_imagines, batch_idx, prefix='train'):
        """The logging half of trainer.py:492-509 (`self.log(f'{prefix}_{key}', value)` per loss term, `-global_step`);
        the TensorBoard visualisation half (`visualise`, trainer.py:569-1020: cv2 / open3d / matplotlib) is out of scope.
        Under Lightning `self.log` is the LightningModule's; in a plain loop the values are handed to `self.log_fn(name,
        value)` if one is set, else kept (still device tensors: no host sync) in `self.logged`."""
        step = getattr(self, 'global_step', 0) if pl is not None else self._global_step
        self.log('-global_step', torch.tensor(-float(step), dtype=torch.float32))
        for key, value in loss.items():
            self.log(f'{prefix}_{key}', value)
        if loss_imagines:
            for key, value in loss_imagines[0].items():
                self.log(f'{prefix}_{key}_imagine', value)

    if pl is None:
        def log(self, name, value, *args, **kwargs):
            if torch.is_tensor(value):
                value = value.detach()
            if self.log_fn is not None:
                self.log_fn(name, value)
            else:
                self.logged[name] = value

    # ------------------------------------------------------------------ data-parallel hooks (Lightning names)
    def on_before_zero_grad(self, optimizer=None):
        pass

    def on_after_backward(self):
        """Lightning calls this right after `loss.backward()`: send the gradient segments backward has not sent itself
        and make the optimizer's stream wait for the exchange.  A plain loop calls it between backward() and step()."""
        if self._reducer is not None:
            if self._reducer.accumulating:
                self._reducer.skip()          # gradients keep accumulating locally; exchanged after the last micro-batch
            else:
                self._reducer.finish()

    def _is_accumulating(self):
        """True for every micro-batch but the last of an ACCUMULATE_GRAD_BATCHES group: Lightning's loop knows
        (`_should_accumulate`); a plain loop sets `self.accumulate_now`."""
        if pl is not None and getattr(self, '_trainer', None) is not None:
            return bool(self._trainer.fit_loop.epoch_loop._should_accumulate())
        return bool(self.accumulate_now)

    def configure_ddp(self, *a, **k):   # pragma: no cover
        raise RuntimeError(_DDP_MSG)

    # ------------------------------------------------------------------ optimiser
    def configure_optimizers(self):
        cfg = self.cfg
        if cfg.OPTIMIZER.FROZEN.ENABLED:
            keep = tuple(cfg.OPTIMIZER.FROZEN.TRAIN_LIST)
            for name, param in self.model.named_parameters():
                if not name.startswith(keep):
                    param.requires_grad = False
        self.store = ParamStore(self.model)
        unused = [p for _, p in self.store.unused]
        optimizer = FusedAdamW(self.store, lr=cfg.OPTIMIZER.LR, weight_decay=cfg.OPTIMIZER.WEIGHT_DECAY,
                               extra_unused=unused)
        if cfg.SCHEDULER.NAME == 'none':
            sched = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda lr: 1)
        elif cfg.SCHEDULER.NAME == 'OneCycleLR':
            sched = torch.optim.lr_scheduler.OneCycleLR(optimizer, max_lr=cfg.OPTIMIZER.LR, total_steps=cfg.STEPS,
                                                        pct_start=cfg.SCHEDULER.PCT_START)
        else:
            raise ValueError(cfg.SCHEDULER.NAME)
        self._optimizer = optimizer
        self._attach_reducer(optimizer)
        return [optimizer], [{'scheduler': sched, 'interval': 'step'}]

    def _attach_reducer(self, optimizer):
        """Data parallelism (reference: Lightning's implicit DDP, train.py:93-98).  When torch.distributed is initialised
        with more than one rank, the segmented reducer (muvo_amd/parallel.py) is attached here: backward hooks launch one
        RCCL all-reduce per finished segment on a side stream, `on_after_backward` waits for them, AdamW divides by the
        world size.  A DistributedDataParallel wrapper around this module would all-reduce a second time (and trip over
        the 12 never-used `encoder_layer.*` tensors): refuse it."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        from muvo_amd.parallel import SegmentedGradReducer
        self._reducer = SegmentedGradReducer(self.store)
        self.model.segment_done = self._reducer.segment_done
        self.model.dropout_rank = dist.get_rank()
        optimizer.grad_scale = self._reducer.grad_scale
        self.register_forward_pre_hook(_refuse_ddp)
