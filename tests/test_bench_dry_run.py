"""CPU: `python bench.py --gpus 8 --dry-run` - the multi-rank control flow of the bench (self-launch through
torch.distributed.run as a child process, RANK / WORLD_SIZE from the environment, replicated weights and per-rank data seeds,
barrier + max-over-ranks timing, the real SegmentedGradReducer with backward hooks over gloo, one JSON line on rank 0) on a toy
model.  The driver launches the real thing as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`; the
reference's counterpart is Lightning's implicit DDP (train.py:93-98)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _run(cmd, env=None):
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize('launcher', ['self', 'driver'])
def test_bench_dry_run_8_ranks(launcher):
    env = dict(os.environ, OMP_NUM_THREADS='1')
    env.pop('WORLD_SIZE', None), env.pop('RANK', None)
    if launcher == 'self':        # bench.py starts the launcher itself
        cmd = [sys.executable, 'bench.py', '--gpus', '8', '--dry-run', '--steps', '3', '--warmup', '1']
    else:                         # the driver's command line
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
               '--master-port', str(29700 + os.getpid() % 200), 'bench.py', '--gpus', '8', '--dry-run', '--steps', '3', '--warmup', '1']
    d = _run(cmd, env)
    assert d['dry_run'] is True and d['n_gpus'] == 8 and d['n_ranks_seen'] == 8 and d['steps'] == 3 and d['warmup'] == 1
    assert d['scaling'] == 'weak' and d['config']['global_batch'] == 16 and d['config']['parallelism'] == 'dp8'
    assert d['value'] > 0 and abs(d['value'] - 16 * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']     # whole-job aggregate
    assert d['per_rank_losses_distinct'] and len(d['per_rank_final_loss']) == 8         # per-rank data seeds
    assert d['params_identical_across_ranks']                                           # replicated init + summed gradients
    ge = d['gradient_exchange']
    assert ge['grad_scale'] == 1.0 / 8
    # every segment of the toy exactly once, in the order backward completes them; all but the last one from a backward hook
    assert ge['order'] == ['rgb_decoder', 'lidar_re', 'voxel_decoder', 'policy', 'rssm', 'fusion', 'lidar_branch', 'image_branch']
    assert list(ge['segments']) == ge['order']
    hooks = [ge['segments'][n]['from_hook'] for n in ge['order']]
    assert hooks[:6] == [True] * 6 and hooks[-1] is False


def test_bench_dry_run_single_rank():
    d = _run([sys.executable, 'bench.py', '--dry-run', '--steps', '2', '--warmup', '1'])
    assert d['n_gpus'] == 1 and d['n_ranks_seen'] == 1 and d['params_identical_across_ranks']
