"""Diagnostic (GPU box): per-tensor deviation of the HIP training step from the committed reference fixture
(tests/golden/base1d_b1s2*), for the conv arithmetic selected by MUVO_CONV_MFMA.  Prints the worst entries of
each family instead of stopping at the first assertion like the pytest does."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import test_model_gpu as T
    dev = torch.device('cuda:0')
    fx, smp, recs = T.run.__wrapped__(dev) if hasattr(T.run, '__wrapped__') else T.run.__pytest_wrapped__.obj(dev)
    g = fx['steps'][0]
    print('losses (rel err):')
    for k, v in g['losses'].items():
        print(f'  {k:18s} {abs(recs[0]["losses"][k] - v) / max(abs(v), 1e-12):.2e}')
    rows = []
    for n, ref in g['grad_l2'].items():
        got = recs[0]['grad_l2'][n]
        if ref is None:
            continue
        rows.append((abs(got - ref) / max(abs(ref), 1e-12), abs(got - ref), n, ref))
    rows.sort(reverse=True)
    print('grad L2 norms, worst 12 (rel, abs, name, ref):')
    for r in rows[:12]:
        print(f'  {r[0]:.2e} {r[1]:.2e} {r[2]} {r[3]:.3e}')
    if 'grad_l2_fp64' in g:
        rows = []
        for n, ref in g['grad_l2_fp64'].items():
            if ref is None:
                continue
            got = recs[0]['grad_l2'][n]
            tol = max(2e-3 * abs(ref), 6.0 * g['grad_l2_ref32_err'][n], 1e-5)
            rows.append((abs(got - ref) / tol, abs(got - ref), ref, g['grad_l2_ref32_err'][n], n))
        rows.sort(reverse=True)
        print('grad L2 norms vs fp64 reference, worst 10 (|err|/tol of the test, |err|, ref norm, reference fp32 err, name):')
        for r in rows[:10]:
            print(f'  {r[0]:.2f} {r[1]:.2e} {r[2]:.3e} {r[3]:.2e} {r[4]}')
    rows = []
    for key in smp.files:
        if key.startswith('grad64.'):
            n = key[7:]
            ref = torch.from_numpy(smp[key]).double()
            ref32 = torch.from_numpy(smp['grad.' + n]).double()
            t = recs[0]['grads'][n].double().contiguous().view(-1)
            stride = max(1, t.numel() // 1024)
            got = t[::stride][:ref.numel()].cpu()
            mx = max(ref.abs().max().item(), 1e-30)
            rows.append(((got - ref).abs().max().item() / mx, (ref32 - ref).abs().max().item() / mx, n))
    rows.sort(reverse=True)
    print('grad samples vs the fp64 reference, max|err|/max|ref|:  ours   reference-fp32   name')
    for r in rows:
        print(f'  {r[0]:.2e} {r[1]:.2e} {r[2]}')
    for step, (rec, gg) in enumerate(zip(recs, fx['steps'])):
        bad = []
        for n, (s_ref, a_ref) in gg['param_checksums_after_step'].items():
            s_got, a_got = rec['checks'][n]
            bad.append((abs(s_got - s_ref) / max(a_ref, 1.0), abs(a_got - a_ref) / max(a_ref, 1e-12), n))
        bad.sort(reverse=True)
        print(f'step {step} param checksum worst 6 (|dsum|/abs-sum, rel abs-sum, name):')
        for r in bad[:6]:
            print(f'  {r[0]:.2e} {r[1]:.2e} {r[2]}')


if __name__ == '__main__':
    main()
