"""Builds libmuvo_hip.so (gfx950) in-tree with hipcc.  `python -m muvo_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmuvo_hip.so')
SOURCES = ['abi.hip', 'conv_gemm.hip', 'conv_vox.hip', 'conv_pw.hip', 'conv_bf3.hip', 'gemm.hip', 'norm.hip', 'elementwise.hip', 'losses.hip', 'metrics.hip', 'bev.hip', 'input.hip', 'augment.hip', 'attention.hip', 'rssm.hip', 'conv_stem.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics', '-Wno-unused-result', '-Wno-unused-value',
         '-ffp-contract=off', '-Rpass-analysis=kernel-resource-usage',
         # A by-value kernel argument struct (ConvPhase, 2.8 KB) is first copied to a private alloca by the front end; InstCombine
         # forwards its loads to the kernel-argument segment only while the alloca has <= this many users (default 300).  The
         # unrolled epilogues (x 6 activations) read `g` more often than that: beyond the limit the whole struct lands in scratch.
         '-mllvm', '-instcombine-max-copied-from-constant-users=8000',
         # No packed fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).  Measured on MI355X (tools/dev/
         # coresidency_repro.py, profiles/r03j_lidar_decoder_stream.txt): `v_pk_mul_f32 d, a, b op_sel:[0,1]` whose b is a register
         # pair filled by ds_read2_b32 returned 0 for the low half in lanes 48-63 while a big-LDS MFMA workgroup of ANOTHER
         # stream was resident on the same compute unit - one product missing from a 1x1 head's sum, in every launch that
         # overlapped, never on an idle GPU or next to other workgroups of the same stream.  The same code with scalar
         # v_mul_f32 / v_add_f32 (this flag) never showed it (0 of 400 launches against 200 of 400); the step time is unchanged
         # (84.5 vs 84.4 ms): nothing here is bound by the fp32 VALU rate.  The flag is a device target feature; the host pass
         # of hipcc reports it as unknown (filtered below).
         ]
NO_PACKED_FP32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
# Translation units that MAY use packed fp32 VALU instructions: the finding above is about a packed product whose operand pair
# came from LDS and is broadcast with op_sel - a VALU dot product against LDS-resident weights.  A unit is listed here only if
# its device code contains NO packed fp32 instruction with an op_sel / op_sel_hi modifier (checked after every build, below):
# conv_gemm.hip (exact-fp32 implicit GEMM: 66 plain v_pk_*_f32 in epilogues and index-free accumulator arithmetic, none with
# op_sel).  gemm.hip (1622 with op_sel: the skinny GEMMs ARE dot products against LDS vectors), conv_vox.hip (486) and
# attention.hip (192) stay without them.
PACKED_FP32_OK = {'conv_gemm.hip'}
# Kernels allowed to use scratch memory (bytes per lane).  Everything else must stay in registers: a kernel that silently
# picks up scratch (an argument struct captured by reference and copied to the stack, register spills after a small edit)
# loses tens of microseconds per workgroup launch — the build fails instead.
SCRATCH_ALLOWED = {'rssm_fwd_kernel': 128, 'rssm_bwd_kernel': 128}


def _check_resources(src, stderr):
    """Parses hipcc's kernel-resource-usage remarks: returns [(kernel, scratch bytes, spilled VGPRs)] of offenders."""
    import re
    bad, name = [], None
    for line in stderr.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r'remark:\s+ScratchSize \[bytes/lane\]: (\d+)', line)
        if m and name and int(m.group(1)) > 0:
            short = next((k for k in SCRATCH_ALLOWED if k in name), None)
            if short is None or int(m.group(1)) > SCRATCH_ALLOWED[short]:
                bad.append((name, int(m.group(1))))
    return bad


def _check_packed_fp32(obj, src_name):
    """Disassembles the gfx950 code object inside `obj` and fails the build if the packed-fp32 workaround is not in effect: no
    v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 at all in a unit built with NO_PACKED_FP32 (a future hipcc that renames or ignores
    the target feature would silently bring them back), none with an op_sel modifier in a unit listed in PACKED_FP32_OK."""
    import re
    import tempfile
    llvm = os.environ.get('MUVO_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, 'dev.co'), os.path.join(td, 'fat.bin')
        r = subprocess.run([os.path.join(llvm, 'llvm-objcopy'), f'--dump-section=.hip_fatbin={fat}', obj], capture_output=True, text=True)
        if r.returncode == 0:
            r = subprocess.run([os.path.join(llvm, 'clang-offload-bundler'), '--type=o', '--unbundle', f'--input={fat}', f'--output={co}',
                                '--targets=hipv4-amdgcn-amd-amdhsa--gfx950'], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            raise RuntimeError(f'packed-fp32 guard: cannot extract the gfx950 code object of {obj}: {r.stderr[-300:]}')
        d = subprocess.run([os.path.join(llvm, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], capture_output=True, text=True)
        if d.returncode != 0:
            raise RuntimeError(f'packed-fp32 guard: llvm-objdump failed on {obj}: {d.stderr[-300:]}')
    pk = [l for l in d.stdout.splitlines() if re.search(r'\bv_pk_(mul|add|fma)_f32\b', l)]
    if src_name in PACKED_FP32_OK:
        bad = [l for l in pk if 'op_sel' in l]
        if bad:
            raise RuntimeError(f'{src_name} is listed in PACKED_FP32_OK but contains {len(bad)} packed fp32 instructions with op_sel '
                               f'(muvo_amd/build.py), e.g. {bad[0].strip()[:120]}')
    elif pk:
        raise RuntimeError(f'{src_name}: {len(pk)} packed fp32 VALU instructions although the unit is built without them '
                           f'(muvo_amd/build.py: NO_PACKED_FP32 no longer takes effect?), e.g. {pk[0].strip()[:120]}')


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_vox.h'), os.path.join(CSRC, 'conv_pw.h'), os.path.join(CSRC, 'conv_plan.h'), os.path.join(CSRC, 'conv_bf3.h'), os.path.join(HERE, '..', 'include', 'muvo_hip.h'),
            os.path.abspath(__file__)]   # the flags are part of the recipe
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            extra = [] if s in PACKED_FP32_OK else NO_PACKED_FP32
            jobs.append([hipcc, *FLAGS, *extra, *os.environ.get('MUVO_HIPCC_EXTRA', '').split(), '-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed:\n{r.stdout}\n{r.stderr}')
        bad = _check_resources(cmd[-3], r.stderr)
        if bad:
            if os.path.exists(cmd[-1]) and cmd[-2] == '-o':
                os.remove(cmd[-1])
            raise RuntimeError('kernels use scratch memory (see muvo_amd/build.py SCRATCH_ALLOWED): ' +
                               ', '.join(f'{n}: {b} B/lane' for n, b in bad))
        rest = '\n'.join(l for l in r.stderr.splitlines() if 'kernel-resource-usage' not in l and 'is not a recognized feature for this target' not in l and not l.lstrip().startswith(('|', '^')) and
                         not (l.strip()[:1].isdigit() and ' | ' in l))
        if verbose and rest.strip():
            print(rest)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    for job in jobs:
        _check_packed_fp32(job[-1], os.path.basename(job[-3]))
    if jobs or force or not os.path.exists(LIB):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs])
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print('built', LIB)
