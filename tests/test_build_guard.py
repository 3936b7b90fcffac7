"""Host logic of the build: hipcc's kernel-resource-usage remarks are parsed and a kernel that uses scratch memory fails the
build (muvo_amd/build.py: a by-value argument struct copied to the stack, or register spills, cost tens of microseconds per
workgroup launch and went unnoticed once)."""
from muvo_amd import build as b

REMARKS = '''
/x/csrc/a.hip:10:1: remark: Function Name: _Z15conv_bf3_kernelILi256ELi128EEv9ConvPhase [-Rpass-analysis=kernel-resource-usage]
/x/csrc/a.hip:10:1: remark:     VGPRs: 175 [-Rpass-analysis=kernel-resource-usage]
/x/csrc/a.hip:10:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
/x/csrc/a.hip:99:1: remark: Function Name: _Z15conv_fwd_kernelILi128ELi128EEv9ConvPhase [-Rpass-analysis=kernel-resource-usage]
/x/csrc/a.hip:99:1: remark:     VGPRs: 108 [-Rpass-analysis=kernel-resource-usage]
/x/csrc/a.hip:99:1: remark:     ScratchSize [bytes/lane]: 2848 [-Rpass-analysis=kernel-resource-usage]
/x/csrc/r.hip:5:1: remark: Function Name: _Z15rssm_fwd_kernel11RssmFwdArgs [-Rpass-analysis=kernel-resource-usage]
/x/csrc/r.hip:5:1: remark:     ScratchSize [bytes/lane]: 64 [-Rpass-analysis=kernel-resource-usage]
'''


def test_scratch_users_are_reported():
    bad = b._check_resources('a.hip', REMARKS)
    assert bad == [('_Z15conv_fwd_kernelILi128ELi128EEv9ConvPhase', 2848)]


def test_allow_list_has_a_size_limit():
    over = REMARKS.replace('ScratchSize [bytes/lane]: 64', 'ScratchSize [bytes/lane]: 4096')
    names = [n for n, _ in b._check_resources('r.hip', over)]
    assert '_Z15rssm_fwd_kernel11RssmFwdArgs' in names


def test_flags_request_the_remarks():
    assert '-Rpass-analysis=kernel-resource-usage' in b.FLAGS and '--offload-arch=gfx950' in b.FLAGS


def test_flags_switch_packed_fp32_off():
    """The library is built without packed fp32 VALU instructions (build.py explains why; the GPU-side check is
    tests/test_kernels_gpu.py::test_head_next_to_a_convolution_of_another_stream), except in the units listed in PACKED_FP32_OK,
    which must not contain the instruction form of the finding (a packed product with an op_sel modifier)."""
    assert b.NO_PACKED_FP32 == ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
    assert b.PACKED_FP32_OK <= {'conv_gemm.hip'}                 # widening the list is a deliberate act: extend this test with the reason
    assert not (b.PACKED_FP32_OK - set(b.SOURCES))


def test_packed_fp32_guard_reads_the_built_objects():
    """The disassembly guard of the build (a future hipcc that ignores the target feature would bring v_pk_*_f32 back unnoticed):
    run over the objects of the in-tree build when they exist, and it must reject a unit that is wrongly declared clean."""
    import os
    objs = [(os.path.join(b.HERE, 'build', s.replace('.hip', '.o')), s) for s in b.SOURCES]
    if not all(os.path.exists(o) for o, _ in objs):
        import pytest
        pytest.skip('library not built in-tree')
    for o, s in objs:
        b._check_packed_fp32(o, s)
    # conv_gemm.o is built WITH packed instructions: declared as a unit without them it must be refused
    if 'conv_gemm.hip' in b.PACKED_FP32_OK:
        import pytest
        with pytest.raises(RuntimeError, match='packed fp32 VALU instructions'):
            b._check_packed_fp32(os.path.join(b.HERE, 'build', 'conv_gemm.o'), 'not_listed.hip')
