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
    tests/test_kernels_gpu.py::test_head_next_to_a_convolution_of_another_stream)."""
    i = b.FLAGS.index('-packed-fp32-ops')
    assert b.FLAGS[i - 3:i] == ['-Xclang', '-target-feature', '-Xclang']
