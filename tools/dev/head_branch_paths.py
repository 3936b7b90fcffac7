"""How often does ops.HeadBranchFn take the in-place accumulate path in one training step?  (ops._grad_is_private)"""
import sys, torch
sys.path.insert(0, '.')
from muvo_amd import ops
from muvo_amd.config import base_1d_cfg
from muvo_amd.data.synthetic import make_batch
from muvo_amd.trainer import WorldModelTrainer
dev = torch.device('cuda:0')
for mode in ('policy', 'f32', 'det'):
    ops.set_conv_mode(ops.CONV_F32 if mode == 'f32' else ops.CONV_BF16X3, min_gflop=-1.0)
    ops.set_deterministic(mode == 'det')
    tr = WorldModelTrainer(base_1d_cfg(RECEPTIVE_FIELD=2, FUTURE_HORIZON=0, STEPS=100).convert_to_dict(), device=dev)
    tr.train()
    seen = []
    real = ops._grad_is_private
    ops._grad_is_private = lambda g: (seen.append(real(g)), seen[-1])[1]
    tr.training_step(make_batch(1, 2, seed=1, device=dev), 0).backward()
    torch.cuda.synchronize()
    ops._grad_is_private = real
    print(mode, 'HeadBranchFn backward calls:', len(seen), 'in place:', sum(seen))
