"""Builds libmuvo_hip.so (gfx950) in-tree with hipcc.  `python -m muvo_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmuvo_hip.so')
SOURCES = ['abi.hip', 'conv_gemm.hip', 'conv_vox.hip', 'conv_pw.hip', 'conv_bf3.hip', 'gemm.hip', 'norm.hip', 'elementwise.hip', 'losses.hip', 'metrics.hip', 'bev.hip', 'input.hip', 'augment.hip', 'attention.hip', 'rssm.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics', '-Wno-unused-result', '-Wno-unused-value',
         '-ffp-contract=off']


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_vox.h'), os.path.join(CSRC, 'conv_pw.h'), os.path.join(CSRC, 'conv_plan.h'), os.path.join(CSRC, 'conv_bf3.h'), os.path.join(HERE, '..', 'include', 'muvo_hip.h')]
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, '-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed:\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs])
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print('built', LIB)
