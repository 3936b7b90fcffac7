"""Mile world model (muvo/models/mile.py:16-161,284-402 construction; :404-593 forward/encode), base_1d branch
(resnet18 image + range-view encoders, DecoderDS, 6-layer transformer fusion, RSSM, policy, RGB / range-view /
voxel decoders), every op on the gfx950 kernels.  Parameter names equal the reference's state_dict."""
import torch
import torch.nn as nn

from muvo_amd import nn as hnn
from muvo_amd import ops
from muvo_amd.layers.layers import BasicBlock
from muvo_amd.bev import FrustumPooling
from muvo_amd.models.common import (BevDecoder, ConvDecoder, Decoder, DecoderDS, Policy, RouteEncode, VoxelDecoder1,
                                    position_embedding_sine)
from muvo_amd.models.resnet import ResNet18Features
from muvo_amd.models.transition import RSSM
from muvo_amd.utils.network_utils import pack_sequence_dim, remove_past, unpack_sequence_dim


class _FeatureConv(nn.Sequential):
    """BasicBlock(s2, downsample) -> BasicBlock -> global avg pool -> flatten (mile.py:104-115)."""

    def __init__(self, cin, cout):
        super().__init__(BasicBlock(cin, cout, stride=2, downsample=True), BasicBlock(cout, cout),
                         hnn.Placeholder(), hnn.Placeholder())

    def forward(self, x):
        return ops.global_avg_pool(self[1](self[0](x, next_convs=(self[1].conv1,))))


class Mile(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.receptive_field = cfg.RECEPTIVE_FIELD
        m = cfg.MODEL
        unsupported = []
        if m.ENCODER.NAME != 'resnet18' or m.LIDAR.ENCODER != 'resnet18':
            unsupported.append('non-resnet18 encoders')
        if not m.TRANSFORMER.ENABLED or m.TRANSFORMER.LARGE:
            unsupported.append('TRANSFORMER.{ENABLED=False,LARGE}')
        if not m.LIDAR.ENABLED or m.LIDAR.POINT_PILLAR.ENABLED:
            unsupported.append('LIDAR off / POINT_PILLAR')
        if m.MEASUREMENTS.ENABLED or m.REWARD.ENABLED or not m.TRANSITION.ENABLED or not m.ROUTE.ENABLED:
            unsupported.append('MEASUREMENTS/REWARD/TRANSITION off/ROUTE off')
        if unsupported:
            raise NotImplementedError('muvo_amd implements the base_1d hot path (SURVEY.md §8); not in scope: '
                                      + ', '.join(unsupported))
        emb, tc = m.EMBEDDING_DIM, m.TRANSFORMER.CHANNELS
        from muvo_amd.config import constant_sizes
        cs_rgb, cs_lidar, cs_voxel = constant_sizes(cfg)     # the reference's (5, 13), (1, 16), (3, 3, 1) unless MODEL.CONSTANT_SIZE says otherwise
        self.encoder = ResNet18Features(3, (2, 3, 4))
        feature_info = self.encoder.feature_info.get_dicts(keys=['num_chs', 'reduction'])
        self.feat_decoder = DecoderDS(feature_info, tc)
        self.bev = bool(m.TRANSFORMER.BEV)
        if self.bev:
            # BEV lifting of the image features (mile.py:33-59)
            self.feat_decoder = Decoder(feature_info, tc)
            bev_downsample = cfg.BEV.FEATURE_DOWNSAMPLE
            self.frustum_pooling = FrustumPooling(
                size=(cfg.BEV.SIZE[0] // bev_downsample, cfg.BEV.SIZE[1] // bev_downsample),
                scale=cfg.BEV.RESOLUTION * bev_downsample, offsetx=cfg.BEV.OFFSET_FORWARD / bev_downsample,
                dbound=cfg.BEV.FRUSTUM_POOL.D_BOUND, downsample=8)
            self.depth_decoder = Decoder(feature_info, tc)
            self.depth = hnn.Conv2d(self.depth_decoder.out_channels, self.frustum_pooling.D, 1)
            self.sparse_depth = cfg.BEV.FRUSTUM_POOL.SPARSE
            self.sparse_depth_count = cfg.BEV.FRUSTUM_POOL.SPARSE_COUNT
            self.bev_down_sample_4 = nn.Sequential(hnn.Conv2d(tc, 512, 5, 2, 2), hnn.Placeholder(), hnn.Conv2d(512, tc, 5, 2, 2))
        self.range_view_encoder = ResNet18Features(4, (2, 3, 4))
        self.range_view_decoder = DecoderDS(self.range_view_encoder.feature_info.get_dicts(keys=['num_chs', 'reduction']), tc)
        self.type_embedding = nn.Parameter(torch.zeros(1, 1, tc, 2))
        # registered but never used in forward, exactly like the reference (mile.py:96-101)
        self.encoder_layer = hnn.TransformerEncoderLayer(tc, 8, dropout=0.1)
        self.transformer_encoder = hnn.TransformerEncoder(tc, 8, num_layers=6, dropout=0.1)
        self.image_feature_conv = _FeatureConv(tc, emb)
        self.lidar_feature_conv = _FeatureConv(tc, emb)
        feature_n_channels = 2 * emb
        self.backbone_route = RouteEncode(m.ROUTE.CHANNELS, m.ROUTE.BACKBONE)
        feature_n_channels += m.ROUTE.CHANNELS
        self.speed_enc = nn.Sequential(hnn.Linear(1, m.SPEED.CHANNELS), hnn.Placeholder(),
                                       hnn.Linear(m.SPEED.CHANNELS, m.SPEED.CHANNELS), hnn.Placeholder())
        feature_n_channels += m.SPEED.CHANNELS
        self.speed_normalisation = cfg.SPEED.NORMALISATION
        self.features_combine = hnn.Linear(feature_n_channels, emb)
        self.rssm = RSSM(embedding_dim=emb, action_dim=m.ACTION_DIM, hidden_state_dim=m.TRANSITION.HIDDEN_STATE_DIM,
                         state_dim=m.TRANSITION.STATE_DIM, action_latent_dim=m.TRANSITION.ACTION_LATENT_DIM,
                         receptive_field=self.receptive_field, use_dropout=m.TRANSITION.USE_DROPOUT,
                         dropout_probability=m.TRANSITION.DROPOUT_PROBABILITY)
        state_dim = m.TRANSITION.HIDDEN_STATE_DIM + m.TRANSITION.STATE_DIM
        self.policy = Policy(in_channels=state_dim)
        if cfg.SEMANTIC_SEG.ENABLED:      # bird's-eye-view semantic + instance segmentation (mile.py:307-313)
            self.bev_decoder = BevDecoder(state_dim, cfg.SEMANTIC_SEG.N_CHANNELS, head='bev')
        if cfg.EVAL.RGB_SUPERVISION:
            self.rgb_decoder = ConvDecoder(state_dim, 3, constant_size=cs_rgb, head='rgb')
        if cfg.LIDAR_RE.ENABLED:
            self.lidar_re = ConvDecoder(state_dim, cfg.LIDAR_RE.N_CHANNELS, constant_size=cs_lidar, head='lidar_re')
        # config-off heads of base_1d that share the ConvDecoder kernels (mile.py:337-363)
        if cfg.LIDAR_SEG.ENABLED:
            self.lidar_segmentation = ConvDecoder(state_dim, cfg.LIDAR_SEG.N_CLASSES, constant_size=cs_lidar, head='lidar_seg')
        if cfg.SEMANTIC_IMAGE.ENABLED:
            self.sem_image_decoder = ConvDecoder(state_dim, cfg.SEMANTIC_IMAGE.N_CLASSES, constant_size=cs_rgb, head='sem_image')
        if cfg.DEPTH.ENABLED:
            self.depth_image_decoder = ConvDecoder(state_dim, 1, constant_size=cs_rgb, head='depth')
        if cfg.VOXEL_SEG.ENABLED:
            self.voxel_decoder = VoxelDecoder1(state_dim, cfg.VOXEL_SEG.N_CLASSES, cfg.VOXEL_SEG.DIMENSION, cs_voxel)
        self._pos_cache = {}
        # data-parallel gradient overlap: called with a segment name (muvo_amd/param_store.SEGMENTS) from autograd
        # hooks the moment backward has finished that segment's parameters
        self.segment_done = None
        # latent memory of deployment_forward / sim_forward (mile.py:399-402)
        self.last_h = self.last_sample = self.last_action = None
        self.count = 0
        self._step_seed = 0
        self.dropout_seed = 0x5EED
        self.dropout_rank = 0     # data-parallel rank (set by WorldModelTrainer.configure_optimizers)
        self.seed_epoch = 0       # optimizer steps taken so far (set by WorldModelTrainer.training_step)

    def _pos(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._pos_cache:
            self._pos_cache[key] = position_embedding_sine(h, w, self.cfg.MODEL.TRANSFORMER.CHANNELS // 2).to(device)
        return self._pos_cache[key]

    def _hook(self, t, segment):
        if self.segment_done is not None and t.requires_grad:
            cb = self.segment_done
            t.register_hook(lambda g, _s=segment: (cb(_s), None)[1])

    def _mark(self, t, segment):
        """Identity on the INPUT of a sub-network: its backward runs after every backward node of that sub-network
        (autograd executes the most recently recorded nodes first), so `segment` and all earlier ones are complete."""
        if self.segment_done is not None and t.requires_grad:
            return ops.segment_mark(t, self.segment_done, segment)
        return t

    def step_seed(self):
        """Dropout seed of this forward: a function of (base seed, data-parallel rank, optimizer steps taken so far —
        restored from a checkpoint —, forwards since construction), so masks differ across ranks and do not repeat
        after a restart."""
        self._step_seed += 1
        return (self.dropout_seed + 0x9E3779B97F4A7C15 * (self.dropout_rank + 1) + 0xBF58476D1CE4E5B9 * self.seed_epoch
                + 0x2545F4914F6CDD1D * self._step_seed) & 0x3FFFFFFFFFFFFFFF

    def forward(self, batch, deployment=False, noise=None, use_prior=None):
        if deployment:
            return self._forward_deployment(batch, use_prior)
        dev = batch['image'].device
        embedding = self.encode(batch)
        b, s = batch['image'].shape[:2]
        action = ops.cat_last([pack_sequence_dim(batch['throttle_brake']), pack_sequence_dim(batch['steering'])])
        action = action.view(b, s, -1)
        self._hook(embedding, 'rssm')
        state_dict = self.rssm(embedding, action, use_sample=True, policy=self.policy, noise=noise, use_prior=use_prior)
        output = {**state_dict}
        post = state_dict['posterior']
        state = ops.cat_last([pack_sequence_dim(post['hidden_state']), pack_sequence_dim(post['sample'])])
        self._hook(state, 'policy')        # d(state) complete: every decoder and the policy are done
        # The decoders only share their input: the voxel decoder runs on a side stream next to the RGB / range-view decoders
        # (ops.branch).  HOST ORDER MATTERS for backward: autograd executes the most recently recorded nodes first and makes the
        # stream of a gradient's CONSUMER wait for its producer the moment the gradient is handed over - every AdaIN of the voxel
        # decoder hands a contribution to d(state), whose consumer lives on the main stream.  Recorded last (the reference's
        # order), the voxel decoder ran its backward first and the main stream queued the RGB decoder's backward behind a wait
        # for the voxel stream's top level; recorded FIRST, its backward is issued after the other decoders' and that wait lands
        # behind them.  Same-box A/B: 82.86 -> 82.45 ms/step (under rocprofv3, where the host is slower than the GPU, the wait
        # showed as 7 ms of main-stream idle time per step: tools/rocpd_queues.py).  The same holds for the range-view decoder
        # (its own stream, ops._STREAM_PLANS): recorded before the RGB decoder.  Record order voxel, range-view, RGB => backward
        # order RGB, range-view, voxel = the order of param_store.SEGMENTS.
        state_ready = ops.stream_event(dev)
        pol = self.policy(state)
        output['throttle_brake'] = unpack_sequence_dim(ops.slice_last(pol, 0, 1), b, s)
        output['steering'] = unpack_sequence_dim(ops.slice_last(pol, 1, 2), b, s)
        joins = []
        if self.cfg.VOXEL_SEG.ENABLED:
            br = ops.branch('decoders', 'voxel_decoder', dev, inputs=(state,), after=state_ready)
            with br:
                output.update(br.out(unpack_sequence_dim(self.voxel_decoder(self._mark(state, 'voxel_decoder')), b, s)))
            joins.append(br)
        if self.cfg.LIDAR_RE.ENABLED:
            br = ops.branch('decoders', 'lidar_decoder', dev, inputs=(state,), after=state_ready)
            with br:
                output.update(br.out(unpack_sequence_dim(self.lidar_re(self._mark(state, 'lidar_re')), b, s)))
            joins.append(br)
        if self.cfg.EVAL.RGB_SUPERVISION:
            output.update(unpack_sequence_dim(self.rgb_decoder(self._mark(state, 'rgb_decoder')), b, s))
        output.update(self._aux_heads(state, b, s))
        for br in joins:
            br.join()
        return output, state_dict

    def _forward_deployment(self, batch, use_prior=None):
        """Mile.forward(batch, deployment=True) (mile.py:404-489): the whole observed sequence through the encoder and the RSSM with the
        recorded `batch['action']` and the distribution MEANS instead of samples (use_sample=False), then everything but the last time
        step is dropped (remove_past, network_utils.py:30-38) and the policy and every decoder run on that last state only (s = 1).
        No caller of the reference uses it (deployment goes through deployment_forward / sim_forward); inference-only here as well."""
        assert self.cfg.MODEL.TRANSITION.ENABLED
        with torch.no_grad():
            embedding = self.encode(batch)
            b, s = batch['image'].shape[:2]
            action = batch['action'].float().contiguous().view(b, s, -1)
            # (in train() mode the RSSM still draws its prior-substitution coin per step, transition.py:118-124, unless `use_prior` says)
            state_dict = self.rssm(embedding, action, use_sample=False, policy=self.policy, use_prior=use_prior)
            state_dict = remove_past(state_dict, s)
            output = {**state_dict}
            post = state_dict['posterior']
            state = ops.cat_last([pack_sequence_dim(post['hidden_state']), pack_sequence_dim(post['sample'])])
            pol = self.policy(state)
            output['throttle_brake'] = unpack_sequence_dim(ops.slice_last(pol, 0, 1), b, 1)
            output['steering'] = unpack_sequence_dim(ops.slice_last(pol, 1, 2), b, 1)
            # reference order (mile.py:448-487); DISPLAY_SEGMENTATION is True (constants.py:4): the BEV decoder runs as well
            if self.cfg.SEMANTIC_SEG.ENABLED:
                output.update(unpack_sequence_dim(self.bev_decoder(state), b, 1))
            if self.cfg.EVAL.RGB_SUPERVISION:
                output.update(unpack_sequence_dim(self.rgb_decoder(state), b, 1))
            if self.cfg.LIDAR_RE.ENABLED:
                output.update(unpack_sequence_dim(self.lidar_re(state), b, 1))
            if self.cfg.LIDAR_SEG.ENABLED:
                output.update(unpack_sequence_dim(self.lidar_segmentation(state), b, 1))
            if self.cfg.SEMANTIC_IMAGE.ENABLED:
                output.update(unpack_sequence_dim(self.sem_image_decoder(state), b, 1))
            if self.cfg.DEPTH.ENABLED:
                output.update(unpack_sequence_dim(self.depth_image_decoder(state), b, 1))
            if self.cfg.VOXEL_SEG.ENABLED:
                output.update(unpack_sequence_dim(self.voxel_decoder(state), b, 1))
        return output, state_dict

    # ------------------------------------------------------------------ closed-loop inference (mile.py:852-1032)
    CARLA_FPS = 10              # constants.py:3

    def _advance_latent(self, batch, action_prev, is_dreaming, frame):
        """encode frame `frame` of the (past-free) batch and advance (last_h, last_sample) one step with `action_prev`"""
        b = batch['image'].shape[0]
        embedding_t = self.encode({k: v[:, frame:frame + 1].contiguous() if frame is not None else v for k, v in batch.items()})[:, -1]
        m = self.cfg.MODEL.TRANSITION
        if self.last_h is None:
            h_t = action_prev.new_zeros(b, m.HIDDEN_STATE_DIM)
            sample_t = action_prev.new_zeros(b, m.STATE_DIM)
        else:
            h_t, sample_t = self.last_h, self.last_sample
        if is_dreaming:
            out = self.rssm.imagine_step(h_t, sample_t, action_prev.contiguous(), use_sample=False, policy=self.policy)
        else:
            out = self.rssm.observe_step(h_t, sample_t, action_prev.contiguous(), embedding_t, use_sample=False,
                                         policy=self.policy)['posterior']
        self.last_h, self.last_sample = out['hidden_state'], out['sample']
        self.count = int(self.CARLA_FPS * self.cfg.DATASET.STRIDE_SEC) - 1

    def _policy_output(self, b):
        state = ops.cat_last([self.last_h, self.last_sample])
        pol = self.policy(state)
        return state, {'throttle_brake': unpack_sequence_dim(ops.slice_last(pol, 0, 1), b, 1),
                       'steering': unpack_sequence_dim(ops.slice_last(pol, 1, 2), b, 1),
                       'hidden_state': self.last_h, 'sample': self.last_sample}

    @torch.no_grad()
    def deployment_forward(self, batch, is_dreaming):
        """mile.py:852-923: keep the latent state between calls; every int(CARLA_FPS * DATASET.STRIDE_SEC) calls the newest
        frame is encoded and the state advanced with batch['action'][:, -2]; the other calls only re-evaluate the policy."""
        assert self.cfg.MODEL.TRANSITION.ENABLED
        b = batch['image'].shape[0]
        if self.count == 0:
            s = batch['image'].shape[1]
            action_t = batch['action'][:, -2]
            cut = {k: v[:, s - 1:].contiguous() for k, v in batch.items()}       # remove_past(batch, s)
            self._advance_latent(cut, action_t, is_dreaming, None)
        else:
            self.count -= 1
        state, output = self._policy_output(b)
        if self.cfg.SEMANTIC_SEG.ENABLED:                                        # constants.DISPLAY_SEGMENTATION = True
            output.update(unpack_sequence_dim(self.bev_decoder(state), b, 1))
        return output

    @torch.no_grad()
    def sim_forward(self, batch, is_dreaming, noise=None):
        """mile.py:925-1032 (sim_run.py:72-73): like deployment_forward, but the state advances with the action of the PREVIOUS
        call, every decoder renders the current state, and the remaining frames of the batch are imagined from it."""
        assert self.cfg.MODEL.TRANSITION.ENABLED
        b = batch['image'].shape[0]
        if self.count == 0:
            s = self.receptive_field
            batch = {k: v[:, s - 1:].contiguous() for k, v in batch.items()}     # remove_past(batch, receptive_field)
            action_t = ops.cat_last([batch['throttle_brake'][:, 0].contiguous(), batch['steering'][:, 0].contiguous()])
            action_last = torch.zeros_like(action_t) if self.last_action is None else self.last_action
            self._advance_latent(batch, action_last, is_dreaming, 0)
            self.last_action = action_t
        else:
            self.count -= 1
        state, output = self._policy_output(b)
        if self.cfg.SEMANTIC_SEG.ENABLED:
            output.update(unpack_sequence_dim(self.bev_decoder(state), b, 1))
        if self.cfg.EVAL.RGB_SUPERVISION:
            output.update(unpack_sequence_dim(self.rgb_decoder(state), b, 1))
        if self.cfg.LIDAR_RE.ENABLED:
            output.update(unpack_sequence_dim(self.lidar_re(state), b, 1))
        output.update(self._aux_heads(state, b, 1))
        if self.cfg.VOXEL_SEG.ENABLED:
            output.update(unpack_sequence_dim(self.voxel_decoder(state), b, 1))
        state_imagine = {'hidden_state': self.last_h, 'sample': self.last_sample, 'throttle_brake': batch['throttle_brake'],
                         'steering': batch['steering']}
        fh = batch['image'].shape[1] - 1
        # (noise: optional explicit (b, fh, S) draws of the imagined steps for parity runs; the reference draws them)
        output_imagine = self.imagine(state_imagine, predict_action=False, future_horizon=fh, noise=noise) if fh > 0 else {}
        return output, output_imagine

    def imagine(self, batch, predict_action=False, future_horizon=None, noise=None):
        """Mile.imagine (mile.py:771-850): roll the prior forward from (hidden_state, sample) with the recorded actions (or
        the policy's) and decode every imagined state.  noise: optional explicit (b, fh, S) draws for parity runs."""
        assert self.cfg.MODEL.TRANSITION.ENABLED
        fh = self.cfg.FUTURE_HORIZON if future_horizon is None else future_horizon
        h_t, sample_t = batch['hidden_state'].contiguous(), batch['sample'].contiguous()
        b = h_t.shape[0]
        tb, st = ops.unstack_time(batch['throttle_brake'].contiguous()), ops.unstack_time(batch['steering'].contiguous())
        out = {'action': [], 'state': [], 'hidden': [], 'sample': []}
        for t in range(fh):
            if predict_action:
                action_t = self.policy(ops.cat_last([h_t, sample_t]))
            else:
                action_t = ops.cat_last([tb[t], st[t]])
            prior_t = self.rssm.imagine_step(h_t, sample_t, action_t, use_sample=True, policy=self.policy,
                                             eps=None if noise is None else noise[:, t])
            sample_t, h_t = prior_t['sample'], prior_t['hidden_state']
            out['action'].append(action_t)
            out['state'].append(ops.cat_last([h_t, sample_t]))
            out['hidden'].append(h_t)
            out['sample'].append(sample_t)
        out = {k: ops.stack_time(v) for k, v in out.items()}
        state = pack_sequence_dim(out['state'])
        pol = self.policy(state)
        out['throttle_brake'] = unpack_sequence_dim(ops.slice_last(pol, 0, 1), b, fh)
        out['steering'] = unpack_sequence_dim(ops.slice_last(pol, 1, 2), b, fh)
        if self.cfg.EVAL.RGB_SUPERVISION:
            out.update(unpack_sequence_dim(self.rgb_decoder(state), b, fh))
        if self.cfg.LIDAR_RE.ENABLED:
            out.update(unpack_sequence_dim(self.lidar_re(state), b, fh))
        out.update(self._aux_heads(state, b, fh))
        if self.cfg.VOXEL_SEG.ENABLED:
            out.update(unpack_sequence_dim(self.voxel_decoder(state), b, fh))
        return out

    def _aux_heads(self, state, b, s):
        out = {}
        if self.cfg.SEMANTIC_SEG.ENABLED:
            out.update(unpack_sequence_dim(self.bev_decoder(self._mark(state, 'bev_decoder')), b, s))
        if self.cfg.LIDAR_SEG.ENABLED:
            out.update(unpack_sequence_dim(self.lidar_segmentation(self._mark(state, 'lidar_segmentation')), b, s))
        if self.cfg.SEMANTIC_IMAGE.ENABLED:
            out.update(unpack_sequence_dim(self.sem_image_decoder(self._mark(state, 'sem_image_decoder')), b, s))
        if self.cfg.DEPTH.ENABLED:
            out.update(unpack_sequence_dim(self.depth_image_decoder(self._mark(state, 'depth_image_decoder')), b, s))
        return out

    def encode(self, batch):
        b, s = batch['image'].shape[:2]
        image = pack_sequence_dim(batch['image'])
        speed = pack_sequence_dim(batch['speed']).contiguous()
        # every branch input is packed BEFORE the inputs-ready event: for a non-contiguous batch tensor (sliced batch, custom
        # collate) pack_sequence_dim is a copy kernel on the main stream, and a side stream that waits only for the event
        # would read the buffer before it is written
        rv_packed = pack_sequence_dim(batch['range_view_pcd_xyzd']).contiguous() if 'range_view_pcd_xyzd' in batch else None
        route_packed = {k: pack_sequence_dim(batch[k]).contiguous() for k in ('route_map',) if k in batch}
        ops.mark_inputs_ready(batch['image'].device)   # packed weights + preprocessed batch are queued: side-stream branches start here
        fc = self.feat_decoder.feat_consumers() if hasattr(self.feat_decoder, 'feat_consumers') else None
        xs = self.encoder(image, feat_consumers=fc)
        x = self.feat_decoder(xs)
        if self.bev:                                                     # mile.py:506-524
            depth = ops.softmax_channel(self.depth(self.depth_decoder(xs)))
            depth_mask = None
            if self.sparse_depth:                                        # only the top-k most likely bins are lifted
                topk_bins = depth.detach().topk(self.sparse_depth_count, dim=1)[1]
                depth_mask = torch.zeros(depth.shape, device=depth.device, dtype=torch.bool)
                depth_mask.scatter_(1, topk_bins, 1)
            x = self.frustum_pooling.lift(x, depth, pack_sequence_dim(batch['intrinsics']).float(),
                                          pack_sequence_dim(batch['extrinsics']).float(), depth_mask)
            x = self.bev_down_sample_4[2](self.bev_down_sample_4[0](x, act=ops.ACT_RELU))
        # recorded before the range-view branch: its backward fires when that branch (and the token gradient) is done
        x = self._mark(x, 'lidar_branch')
        rv = rv_packed
        br_lidar = ops.branch('lidar', 'lidar_encoder', x.device, inputs=(rv,))
        with br_lidar:                 # next to the image encoder's kernels still queued on the main stream
            lidar_features = br_lidar.out(self.range_view_decoder(
                self.range_view_encoder(rv, feat_consumers=self.range_view_decoder.feat_consumers())))
        br_lidar.join()
        hi, wi = x.shape[-2:]
        hl, wl = lidar_features.shape[-2:]
        # x + pos -> flatten/permute -> + type embedding -> concat (mile.py:542-557), one transpose kernel per sensor
        tokens = ops.make_tokens(x, lidar_features, self._pos(hi, wi, x.device), self._pos(hl, wl, x.device),
                                 self.type_embedding)
        self._hook(tokens, 'fusion')
        tokens_out = self.transformer_encoder(tokens, self.step_seed())
        image_tokens_out = ops.untoken(tokens_out, 0, hi, wi)
        lidar_tokens_out = ops.untoken(tokens_out, hi * wi, hl, wl)
        # route-map encoder (ResNet-18 on 64 x 64 pixels) and speed encoder: ~300 launches of 5-50 us that occupy a few compute
        # units - on a side stream, next to whatever the main stream has queued (they depend on the preprocessed batch only)
        route_map = route_packed['route_map']
        br_route = ops.branch('route', 'route_encoder', x.device, inputs=(route_map, speed))
        with br_route:
            route_features = br_route.out(self.backbone_route(route_map))
            sp = ops.divide_scalar(speed.float(), self.speed_normalisation)
            sp = br_route.out(self.speed_enc[2](self.speed_enc[0](sp, act=ops.ACT_RELU), act=ops.ACT_RELU))
        features = [self.image_feature_conv(image_tokens_out), self.lidar_feature_conv(lidar_tokens_out)]
        br_route.join()
        features += [route_features, sp]
        embedding = self.features_combine(ops.cat_last(features))
        return unpack_sequence_dim(embedding, b, s)
