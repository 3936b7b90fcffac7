import sys, torch
sys.path.insert(0, '.')
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch, make_noise
from muvo_amd.trainer import WorldModelTrainer
from muvo_amd.utils import detinit
dev = torch.device('cuda:0')
b, s = 1, 2
tr = WorldModelTrainer(base_1d_cfg(RECEPTIVE_FIELD=s, FUTURE_HORIZON=0, STEPS=100000).convert_to_dict(), device=dev)
tr.train(); tr.preprocess.augment = False
detinit.fill_state_dict_(tr.model)
for layer in tr.model.transformer_encoder.layers:
    layer.p = 0.0
opt = tr.configure_optimizers()[0][0]
eps, use_prior = make_noise(b, s, seed=1234); eps = eps.to(dev)
state = {k: v.clone() for k, v in tr.model.state_dict().items()}
outs = []
for name, st in (('off', False), ('on-1', True), ('on-2', True), ('off-2', False), ('on-3', True)):
    ops.STREAMS = st
    tr.model.load_state_dict(state)
    opt.zero_grad()
    batch = make_batch(b, s, seed=1234, device=dev)
    losses, output, _, _ = tr.shared_step(batch, mode='train', noise=eps, use_prior=use_prior)
    tr.loss_reducing(losses).backward()
    torch.cuda.synchronize()
    outs.append((name, {k: v.detach().clone() for k, v in output.items() if torch.is_tensor(v) and k.startswith('voxel')},
                 {k: v.item() for k, v in losses.items() if k.startswith('voxel')}))
ref = outs[0]
for name, o, l in outs:
    d = {k: ((o[k] - ref[1][k]).abs().max() / ref[1][k].abs().max()).item() for k in o}
    print(name, {k: f'{v:.2e}' for k, v in d.items()}, l)
