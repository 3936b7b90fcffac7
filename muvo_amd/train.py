"""Thin counterpart of the reference's `train.py` (train.py:51-115) for the MI355X step: same command line
(`muvo/config.py:326-369`: `--config-file`, trailing `opts`), the plain loop that Lightning's automatic optimisation runs on a
LightningModule — training_step -> zero_grad -> backward -> (gradient exchange) -> optimizer.step -> scheduler.step with
`OPTIMIZER.ACCUMULATE_GRAD_BATCHES` micro-batches per step — the per-step loss log (trainer.py:492-499) and the checkpoint
callback: every `VAL_CHECK_INTERVAL` steps a Lightning-format checkpoint is written INTO THE CURRENT WORKING DIRECTORY under
Lightning's file name (`MyModelCheckpoint`, train.py:31-48: `filename = filepath.split('/')[-1]`).

The dataset (CARLA recordings, muvo/data) is out of scope: batches are synthetic with the reference's batch schema
(`muvo_amd/data/synthetic.py`).  One process per GPU; under `torch.distributed.run` the ranks form an RCCL group and
`WorldModelTrainer` exchanges gradients itself (muvo_amd/parallel.py) — no DistributedDataParallel wrapper.

    python -m muvo_amd.train --config-file muvo_amd/configs/test_base_1d.yml STEPS 100 BATCHSIZE 2 [--resume epoch=0-step=50.ckpt]
"""
import json
import os
import sys
import time

import torch

from muvo_amd.config import get_cfg, get_parser
from muvo_amd.data.synthetic import make_batch
from muvo_amd.trainer import WorldModelTrainer


def checkpoint_dict(module, optimizer, scheduler, global_step):
    """What Lightning's `dump_checkpoint` writes for this module (the keys the reference reads back: `state_dict` with the
    `model.` prefix, trainer.py:202-211; optimizer / scheduler states for resuming)."""
    return {'epoch': 0, 'global_step': global_step, 'pytorch-lightning_version': 'muvo_amd-plain-loop',
            'state_dict': module.state_dict(), 'optimizer_states': [optimizer.state_dict()],
            'lr_schedulers': [scheduler.state_dict()], 'hyper_parameters': {'hparams': module.cfg.convert_to_dict()},
            'world_size': int(os.environ.get('WORLD_SIZE', '1'))}


def save_checkpoint(module, optimizer, scheduler, global_step, save_dir):
    filepath = os.path.join(save_dir, f'epoch=0-step={global_step}.ckpt')
    filename = filepath.split('/')[-1]            # train.py:33: the file lands in the current working directory
    torch.save(checkpoint_dict(module, optimizer, scheduler, global_step), filename)
    return filename


def load_checkpoint(module, optimizer, scheduler, path):
    ck = torch.load(path, map_location='cpu', weights_only=False)
    # like load_pretrained_weights (trainer.py:202-211): the `model.` entries, strict; whatever else a Lightning checkpoint of
    # the reference carries in `state_dict` (loss-module buffers) is not needed
    module.model.load_state_dict({k[6:]: v for k, v in ck['state_dict'].items() if k[:6] == 'model.'}, strict=True)
    optimizer.load_state_dict(ck['optimizer_states'][0])
    scheduler.load_state_dict(ck['lr_schedulers'][0])
    return int(ck['global_step'])


def fit(cfg, device, steps=None, resume=None, log=None, seed=1234, batch_fn=None, setup=None):
    """The training loop; returns (module, list of per-step loss dicts as floats).  batch_fn(micro_index) / setup(module):
    hooks for tests (own batches, e.g. switching dropout off)."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    torch.manual_seed(seed)
    module = WorldModelTrainer(cfg.convert_to_dict(), device=device)
    module.train()
    if setup is not None:
        setup(module)
    opts, scheds = module.configure_optimizers()
    optimizer, scheduler = opts[0], scheds[0]['scheduler']
    torch.manual_seed(seed + 7919 * rank)         # data-dependent randomness (RSSM noise, augmentation) differs per rank
    global_step = load_checkpoint(module, optimizer, scheduler, resume) if resume else 0
    module._global_step = global_step
    steps = cfg.STEPS if steps is None else steps
    accum = max(1, int(cfg.OPTIMIZER.ACCUMULATE_GRAD_BATCHES))
    s = cfg.RECEPTIVE_FIELD + cfg.FUTURE_HORIZON
    world = dist.get_world_size() if dist.is_initialized() else 1
    history, micro = [], global_step * accum
    t0 = time.time()
    while global_step < steps:
        # every optimizer step starts from its own seed: a resumed run draws the same RSSM noise / augmentation as the
        # uninterrupted one (the dropout seeds already depend on (rank, optimizer step, micro-batch) only)
        torch.manual_seed(seed + 7919 * rank + 104729 * (global_step + 1))
        optimizer.zero_grad()
        for k in range(accum):
            module.accumulate_now = k < accum - 1
            batch = batch_fn(micro) if batch_fn else make_batch(cfg.BATCHSIZE, s, seed=seed + micro * world + rank, device=device)
            loss = module.training_step(batch, micro)
            (loss / accum if accum > 1 else loss).backward()      # Lightning divides the loss by accumulate_grad_batches
            module.on_after_backward()
            micro += 1
        optimizer.step()
        scheduler.step()
        global_step += 1
        module._global_step = global_step
        if log is not None or global_step % max(1, cfg.LOGGING_INTERVAL) == 0 or global_step == steps:
            rec = {k: float(v.detach()) for k, v in module.logged.items()}
            rec['step'], rec['lr'] = global_step, optimizer.param_groups[0]['lr']
            history.append(rec)
            if rank == 0:
                total = sum(v for k, v in rec.items() if k.startswith('train_'))
                line = json.dumps({'step': global_step, 'loss': total, 'lr': rec['lr'], 's_per_step': (time.time() - t0) / len(history)})
                (log or print)(line)
        if cfg.VAL_CHECK_INTERVAL and global_step % cfg.VAL_CHECK_INTERVAL == 0 and rank == 0:
            name = save_checkpoint(module, optimizer, scheduler, global_step, cfg.LOG_DIR)
            (log or print)(f'checkpoint {name}')
    return module, history


def main(argv=None):
    parser = get_parser()
    parser.add_argument('--resume', default='', help='Lightning-format checkpoint to continue from')
    args = parser.parse_args(argv)
    cfg = get_cfg(args)
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=device)
    fit(cfg, device, resume=args.resume or None)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    sys.exit(main())
