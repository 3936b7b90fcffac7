"""PreProcess (muvo/models/preprocess.py:13-225) as fused HIP kernels, base_1d inputs.

Same contract as the reference: mutates and returns the caller's batch dict, adding rgb_label_{1,2,4},
range_view_label_{1,2,4}, voxel_label_{1,2,4}; `rgb_label_1` is the cropped [0,1] image (pre-normalisation) and
`range_view_label_1` IS the range-view network input.  Pixel/route augmentation
(preprocess.py:295-367, training only, torchvision) is not part of this round's path (next rows)."""
import torch
import torch.nn as nn

from muvo_amd import ops


class PreProcess(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.crop = tuple(cfg.IMAGE.CROP)
        self.route_map_size = cfg.ROUTE.SIZE
        self.mean = tuple(float(v) for v in cfg.IMAGE.IMAGENET_MEAN)
        self.std = tuple(float(v) for v in cfg.IMAGE.IMAGENET_STD)

    def forward(self, batch):
        cfg = self.cfg
        # /255, crop, label pyramid, ImageNet normalisation (preprocess.py:203-218, :102-113)
        label1, image = ops.preprocess_image(batch['image'], self.crop, self.mean, self.std)
        batch['image'] = image
        if 'route_map' in batch:
            batch['route_map'] = ops.preprocess_route(batch['route_map'], self.route_map_size, self.mean, self.std)
        if 'intrinsics' in batch:
            intr = batch['intrinsics'].clone()
            intr[..., 0, 2] -= self.crop[0]
            intr[..., 1, 2] -= self.crop[1]
            batch['intrinsics'] = intr
        if cfg.EVAL.RGB_SUPERVISION:
            batch['rgb_label_1'] = label1
            h, w = label1.shape[-2:]
            for f in (2, 4):
                batch[f'rgb_label_{f}'] = ops.resize_bilinear(batch[f'rgb_label_{f // 2}'], h // f, w // f)
        if cfg.LIDAR_RE.ENABLED:
            rv = ops.divide_scalar(batch['range_view_pcd_xyzd'].float(), cfg.LIDAR_RE.SCALE)
            batch['range_view_pcd_xyzd'] = rv
            batch['range_view_label_1'] = rv
            h, w = rv.shape[-2:]
            for f in (2, 4):
                batch[f'range_view_label_{f}'] = ops.resize_nearest(batch[f'range_view_label_{f // 2}'], (h // f, w // f))
        # bird's-eye-view labels (preprocess.py:50-100; EVAL.MASK_VIEW off): rotate 90 degrees clockwise, nearest pyramids,
        # instance ids -> centre heat map + offsets at every scale (sigma / scale)
        if cfg.SEMANTIC_SEG.ENABLED and 'birdview_label' in batch:
            bev = torch.rot90(batch['birdview_label'], k=-1, dims=[3, 4]).to(torch.uint8).contiguous()
            batch['birdview_label'] = bev
            batch['birdview_label_1'] = bev
            h, w = bev.shape[-2:]
            for f in (2, 4):
                batch[f'birdview_label_{f}'] = ops.resize_nearest(batch[f'birdview_label_{f // 2}'], (h // f, w // f))
        if cfg.SEMANTIC_SEG.ENABLED and 'instance_label' in batch:
            inst = torch.rot90(batch['instance_label'], k=-1, dims=[3, 4]).to(torch.uint8).contiguous()
            batch['instance_label'] = inst
            sigma, ign = cfg.INSTANCE_SEG.CENTER_LABEL_SIGMA_PX, cfg.INSTANCE_SEG.IGNORE_INDEX
            batch['center_label'], batch['offset_label'] = ops.instance_labels(inst, sigma, ign)
            batch['instance_label_1'], batch['center_label_1'], batch['offset_label_1'] = inst, batch['center_label'], batch['offset_label']
            h, w = inst.shape[-2:]
            for f in (2, 4):
                batch[f'instance_label_{f}'] = ops.resize_nearest(batch[f'instance_label_{f // 2}'], (h // f, w // f))
                batch[f'center_label_{f}'], batch[f'offset_label_{f}'] = ops.instance_labels(batch[f'instance_label_{f}'], sigma / f, ign)
        # config-off inputs of base_1d (preprocess.py:127-149,164-175,228-241): crop like the image, then label pyramids
        left, top, right, bottom = self.crop
        if cfg.SEMANTIC_IMAGE.ENABLED:
            sem = batch['semantic_image'][..., top:bottom, left:right].contiguous()
            batch['semantic_image'] = sem
            batch['semantic_image_label_1'] = sem
            h, w = sem.shape[-2:]
            for f in (2, 4):
                batch[f'semantic_image_label_{f}'] = ops.resize_nearest(batch[f'semantic_image_label_{f // 2}'].to(torch.uint8),
                                                                        (h // f, w // f))
        if cfg.DEPTH.ENABLED:
            dep = batch['depth'][..., top:bottom, left:right].float().contiguous()
            batch['depth'] = dep
            batch['depth_label_1'] = dep
            h, w = dep.shape[-2:]
            for f in (2, 4):
                batch[f'depth_label_{f}'] = ops.resize_bilinear(batch[f'depth_label_{f // 2}'], h // f, w // f)
        if cfg.LIDAR_SEG.ENABLED:
            seg = batch['range_view_pcd_seg'].to(torch.uint8).contiguous()
            batch['range_view_seg_label_1'] = seg
            h, w = seg.shape[-2:]
            for f in (2, 4):
                batch[f'range_view_seg_label_{f}'] = ops.resize_nearest(batch[f'range_view_seg_label_{f // 2}'], (h // f, w // f))
        if cfg.VOXEL_SEG.ENABLED:
            batch['voxel_label_1'] = batch['voxel']
            x, y, z = batch['voxel'].shape[-3:]
            for f in (2, 4):
                batch[f'voxel_label_{f}'] = ops.resize_nearest(batch[f'voxel_label_{f // 2}'], (x // f, y // f, z // f))
        return batch
