"""PreProcess (muvo/models/preprocess.py:13-225) as fused HIP kernels, base_1d inputs.

Same contract as the reference: mutates and returns the caller's batch dict, adding rgb_label_{1,2,4},
range_view_label_{1,2,4}, voxel_label_{1,2,4}; `rgb_label_1` is the cropped [0,1] image (pre-normalisation) and
`range_view_label_1` IS the range-view network input.  In training mode the pixel / route augmentation
of the reference (preprocess.py:45-48,213-214,295-367) runs too: the random draws are made on the host in the reference's
RNG call order (muvo_amd/augment.py) — or passed in as explicit tables (`batch['_pixel_aug']`, `batch['_route_aug']`) —
and applied by csrc/augment.hip.  Like in the reference the pixel augmentation alters `rgb_label_1` (it IS the image) but
not `rgb_label_2/4`, which are made before it (preprocess.py:104-113 run before :213-214)."""
import torch
import torch.nn as nn

from muvo_amd import ops


def bev_out_of_view_mask(fov, image_width, resolution, crop_left, crop_right, bev_w, bev_h, offset_forward, camera_forward):
    """(bev_h, bev_w) bool: True = this bird's-eye-view cell is outside the camera's horizontal field of view or behind the ego
    vehicle (EVAL.MASK_VIEW; muvo/utils/geometry_utils.py:37-61).  Column u of a ground point (x right, z forward) through a
    pinhole with the principal point moved by the crop: u = x / z * f + c_u; visible when 0 <= u < cropped width.  Rows run
    from the far end of the grid towards the vehicle; the rows between camera and grid end are all masked."""
    import numpy as np
    f = image_width / (2 * np.tan(fov * np.pi / 360.0))
    c_u = image_width / 2 - crop_left
    half = np.round((bev_w // 2) * resolution, decimals=1)
    cam_off = (bev_h / 2 + offset_forward) * resolution + camera_forward
    top = np.round(bev_h * resolution - cam_off, decimals=1)
    x, z = np.arange(-half, half, resolution), np.arange(0.01, top, resolution)
    u = x / z[:, None] * f + c_u
    visible = (u >= 0) & (u < crop_right - crop_left)
    behind = np.ones((int(cam_off / resolution), visible.shape[1]), dtype=bool)
    return np.vstack([~visible[::-1], behind])


class PreProcess(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.crop = tuple(cfg.IMAGE.CROP)
        self.route_map_size = cfg.ROUTE.SIZE
        self.mean = tuple(float(v) for v in cfg.IMAGE.IMAGENET_MEAN)
        self.std = tuple(float(v) for v in cfg.IMAGE.IMAGENET_STD)
        # EVAL.RESOLUTION (preprocess.py:209-210): antialiased down-scaling of the cropped image by 1 / FACTOR before the model sees
        # it (the reference's RGB losses then compare 320 x 832 predictions with the smaller label and fail to broadcast: the
        # switch only works with EVAL.RGB_SUPERVISION off, here as there)
        self.eval_scale = 1.0 / cfg.EVAL.RESOLUTION.FACTOR if cfg.EVAL.RESOLUTION.ENABLED else None
        if self.eval_scale is not None and self.eval_scale != 1.0 and cfg.EVAL.RGB_SUPERVISION:
            raise ValueError('EVAL.RESOLUTION with FACTOR != 1 needs EVAL.RGB_SUPERVISION = False: the RGB decoder reconstructs the '
                             'full-size image while rgb_label_1 is the down-scaled one (the reference fails in its loss, trainer.py:296-302)')
        self._pins = {}
        self.bev_out_of_view_mask = None
        if cfg.EVAL.MASK_VIEW:        # preprocess.py:20-21
            self.bev_out_of_view_mask = torch.from_numpy(bev_out_of_view_mask(
                cfg.IMAGE.FOV, cfg.IMAGE.SIZE[1], cfg.BEV.RESOLUTION, cfg.IMAGE.CROP[0], cfg.IMAGE.CROP[2], cfg.BEV.SIZE[0],
                cfg.BEV.SIZE[1], cfg.BEV.OFFSET_FORWARD, cfg.IMAGE.CAMERA_POSITION[0]))
        # state_dict parity with the reference module (preprocess.py:42-43); the kernels take the values as scalars
        self.register_buffer('image_mean', torch.tensor(self.mean).unsqueeze(1).unsqueeze(1))
        self.register_buffer('image_std', torch.tensor(self.std).unsqueeze(1).unsqueeze(1))
        self.augment = True     # False: no augmentation even in training mode (parity runs against augmentation-free fixtures)

    def _to_device(self, table, device):
        """Small host table -> device through a ring of four pinned buffers PER TABLE SHAPE.  Each slot remembers the event
        recorded behind its last host-to-device copy and waits for it before the slot is rewritten: the training loop never
        synchronises, so the host runs several steps ahead of the stream and would otherwise overwrite a table whose copy
        has not executed yet (a torn or future table)."""
        key = tuple(table.shape)
        ring = self._pins.get(key)
        if ring is None:
            ring = self._pins[key] = dict(i=0, buf=[torch.empty(table.shape, dtype=torch.float32).pin_memory() for _ in range(4)],
                                          ev=[None] * 4)
        i = ring['i'] = (ring['i'] + 1) % 4
        if ring['ev'][i] is not None:
            ring['ev'][i].synchronize()
        ring['buf'][i].copy_(table)
        out = ring['buf'][i].to(device, non_blocking=True)
        if out.is_cuda:
            ring['ev'][i] = torch.cuda.Event()
            ring['ev'][i].record(torch.cuda.current_stream(out.device))
        return out

    def forward(self, batch):
        cfg = self.cfg
        # /255, crop, label pyramid, ImageNet normalisation (preprocess.py:203-218, :102-113)
        b, s = batch['image'].shape[:2]
        dev = batch['image'].device
        pix_aug = batch.pop('_pixel_aug', None)
        route_aug = batch.pop('_route_aug', None)
        if self.training and self.augment and pix_aug is None:          # same RNG call order as the reference: pixels first, then routes
            from muvo_amd import augment
            pix_aug = augment.draw_pixel_params(cfg, b, s)
            if 'route_map' in batch and route_aug is None:
                route_aug = augment.draw_route_params(cfg, b, self.route_map_size)
        label1, image = ops.preprocess_image(batch['image'], self.crop, self.mean, self.std)
        scale = self.eval_scale
        if scale is not None and scale != 1.0:          # functional_resize_batch (preprocess.py:252-273)
            h, w = label1.shape[-2:]
            label1, image = ops.resize_bilinear_aa(label1, int(round(h * scale)), int(round(w * scale)), self.mean, self.std)
            for k in ('image_instance_mask', 'semantic_image'):
                if k in batch:
                    raise NotImplementedError(f'EVAL.RESOLUTION: antialiased resize of `{k}` is not built')
        batch['image'] = image
        if 'route_map' in batch:
            if route_aug is not None and bool((route_aug[:, 0] != 0).any()):
                route_aug = route_aug if route_aug.is_cuda else self._to_device(route_aug, dev)
                batch['route_map'] = ops.preprocess_route_aug(batch['route_map'], self.route_map_size, self.mean, self.std, route_aug)
            else:
                batch['route_map'] = ops.preprocess_route(batch['route_map'], self.route_map_size, self.mean, self.std)
        if 'intrinsics' in batch:
            intr = batch['intrinsics'].clone()
            intr[..., 0, 2] -= self.crop[0]
            intr[..., 1, 2] -= self.crop[1]
            if scale is not None and scale != 1.0:
                intr[..., :2, :] *= scale
            batch['intrinsics'] = intr
        if cfg.EVAL.RGB_SUPERVISION:
            batch['rgb_label_1'] = label1
            h, w = label1.shape[-2:]
            for f in (2, 4):
                batch[f'rgb_label_{f}'] = ops.resize_bilinear(batch[f'rgb_label_{f // 2}'], h // f, w // f)
        if cfg.LOSSES.RGB_INSTANCE and 'image_instance_mask' in batch:     # preprocess.py:115-125,242-243: crop, nearest pyramid
            left, top, right, bottom = self.crop
            im = batch['image_instance_mask'][..., top:bottom, left:right].to(torch.uint8).contiguous()
            batch['image_instance_mask'] = batch['image_instance_mask_1'] = im
            h, w = im.shape[-2:]
            for f in (2, 4):
                batch[f'image_instance_mask_{f}'] = ops.resize_nearest(batch[f'image_instance_mask_{f // 2}'], (h // f, w // f))
        # PixelAugmentation: after the label pyramid, in place on the [0,1] image = rgb_label_1, re-normalising `image`
        if pix_aug is not None and bool(((pix_aug[:, 0] != 0) | (pix_aug[:, 2] != 0)).any()):
            pix_aug = pix_aug if pix_aug.is_cuda else self._to_device(pix_aug, dev)
            ops.pixel_augment(label1, image, pix_aug, self.mean, self.std)
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
            if self.bev_out_of_view_mask is not None:      # preprocess.py:52-54: cells the camera cannot see -> class 0, in place
                batch['birdview_label'][:, :, :, self.bev_out_of_view_mask.to(batch['birdview_label'].device)] = 0
            bev = torch.rot90(batch['birdview_label'], k=-1, dims=[3, 4]).to(torch.uint8).contiguous()
            batch['birdview_label'] = bev
            batch['birdview_label_1'] = bev
            h, w = bev.shape[-2:]
            for f in (2, 4):
                batch[f'birdview_label_{f}'] = ops.resize_nearest(batch[f'birdview_label_{f // 2}'], (h // f, w // f))
        if cfg.SEMANTIC_SEG.ENABLED and 'instance_label' in batch:
            if self.bev_out_of_view_mask is not None:      # preprocess.py:70-72
                batch['instance_label'][:, :, :, self.bev_out_of_view_mask.to(batch['instance_label'].device)] = 0
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
