"""CPU-only checks of the C-ABI boundary: the library loads, exports every symbol include/w2s.h declares, the header is
valid C, and the ctypes structs have the C layout.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'w2s.h')


@pytest.fixture(scope='module')
def built_lib():
    from wav2sleep_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    return lib


def declared_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r'\b(w2s_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol(built_lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    dll = ctypes.CDLL(built_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(dll, s)]
    assert not missing, missing
    assert set(built_lib.EXPORTS) <= set(syms), set(built_lib.EXPORTS) - set(syms)
    dll.w2s_version.restype = ctypes.c_char_p
    assert b'gfx950' in dll.w2s_version()


def test_header_is_plain_c_and_struct_layout_matches_ctypes(built_lib, tmp_path):
    src = tmp_path / 't.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "w2s.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(w2s_conv_args), offsetof(w2s_conv_args, B), offsetof(w2s_conv_args, epi), sizeof(w2s_wgrad_args), '
                   'offsetof(w2s_wgrad_args, B), offsetof(w2s_wgrad_args, nslab), sizeof(w2s_reduce_job), offsetof(w2s_reduce_job, layout), '
                   'sizeof(w2s_repack_job), offsetof(w2s_repack_job, taps), sizeof(w2s_colsum_job), offsetof(w2s_colsum_job, accumulate));'
                   'return 0;}\n')
    exe = tmp_path / 't'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    L = built_lib
    want = [ctypes.sizeof(L.ConvArgs), L.ConvArgs.B.offset, L.ConvArgs.epi.offset, ctypes.sizeof(L.WgradArgs), L.WgradArgs.B.offset,
            L.WgradArgs.nslab.offset, ctypes.sizeof(L.ReduceJob), L.ReduceJob.layout.offset, ctypes.sizeof(L.RepackJob), L.RepackJob.taps.offset,
            ctypes.sizeof(L.ColsumJob), L.ColsumJob.accumulate.offset]
    assert [int(v) for v in out] == want


def test_argument_validation_without_gpu(built_lib):
    """Entry points reject bad descriptors before touching the device (EINVAL = -1), so this is safe on CPU."""
    L = built_lib
    dll = L.load()
    a = L.ConvArgs()
    assert dll.w2s_conv_forward(ctypes.byref(a), None) == -1
    assert dll.w2s_conv_forward(None, None) == -1
    w = L.WgradArgs()
    assert dll.w2s_wgrad(ctypes.byref(w), None) == -1
    assert dll.w2s_stats_finalize(None, 1, 1, 16, ctypes.c_long(1), ctypes.c_float(0.01), 0, None, None) == -1
    assert L.conv_tile(16, 16, 3, 1) == 256 and L.conv_tile(128, 128, 3, 1) == 64
    assert L.wgrad_grid_y(128, 128, 3) == 1 and L.wgrad_grid_y(128, 128, 7, 2) == 7 and L.wgrad_grid_y(16, 16, 3) == 1
    assert L.wgrad_slabs_per_block(128, 128, 3) == 1 and L.wgrad_slabs_per_block(16, 16, 3) == 4 and L.bwd_fused_tile(16, 16) == 254 and L.bwd_fused_tile(16, 16, 1, True) == 126 and L.bwd_fused_tile(16, 16, 1, False, False) == 256
    assert L.conv_fwd_fused_tile(16, 16, 1) == 254 and L.conv_fwd_fused_tile(32, 32, 2) == 127 and L.conv_fwd_fused_tile(64, 64, 1) == 0
    assert L.bwd_fused_folds_residual(16, 16) and L.bwd_fused_folds_residual(32, 16) and not L.bwd_fused_folds_residual(32, 32)
    assert dll.w2s_wgrad_reduce_batch(None, 1, None) == -1 and dll.w2s_repack_batch(None, 0, None) == -1 and dll.w2s_colsum_batch(None, 3, None) == -1
    assert dll.w2s_conv_fwd_fused(None, None, None, None, None, None, 1, 8, 8, 16, 16, 1, 2, 4, None) == -1
    # transformer epilogue fusions (W2S_FUSE_* in `reserved`): a bit whose operand is NULL, or any bit outside the bias epilogue, is EINVAL
    # (the pointers are never dereferenced on this path: validation happens before the launch)
    def desc(**kw):
        d = L.ConvArgs()
        d.x = d.w = d.y = 4096
        d.B, d.L_in, d.L_out, d.cin, d.cout, d.taps, d.stride, d.dil, d.ldx, d.ldy, d.epi = 1, 64, 64, 128, 128, 1, 1, 1, 128, 128, L.EPI_BIAS
        for k, v in kw.items():
            setattr(d, k, v)
        return d
    assert dll.w2s_conv_forward(ctypes.byref(desc(reserved=L.FUSE_ADD_DROP)), None) == -1              # aux == NULL
    assert dll.w2s_conv_forward(ctypes.byref(desc(reserved=L.FUSE_GELU_BWD_DROP)), None) == -1         # aux == NULL
    assert dll.w2s_conv_forward(ctypes.byref(desc(reserved=L.FUSE_Y2_GELU_DROP)), None) == -1          # y2 == NULL
    assert dll.w2s_conv_forward(ctypes.byref(desc(reserved=L.FUSE_ADD_DROP, aux=4096, epi=L.EPI_PLAIN)), None) == -1   # not the bias epilogue


def test_cpu_tensors_are_refused(built_lib):
    import torch
    with pytest.raises(built_lib.W2SError):
        built_lib.eltwise(0, torch.zeros(4), None, torch.zeros(4), 4)


def test_library_has_no_packed_fp32_instruction_with_src1_high_half_in_the_low_lane():
    """gfx950 erratum found in round 3 (tools/pk_fma_opsel_repro.hip): v_pk_{fma,mul,add}_f32 with op_sel:[x,1,..] return wrong low-lane
    results while a bf16 MFMA runs on the same CU.  The shipped code objects must not contain that form (wav2sleep_amd/isa_audit.py)."""
    import os
    import pytest
    from wav2sleep_amd import isa_audit, lib
    if not os.path.exists(isa_audit.OBJDUMP):
        pytest.skip('llvm-objdump not in this image')
    n, bad = isa_audit.audit(lib.LIB_PATH)
    assert n > 100000, n          # the disassembly really saw the kernels
    assert not bad, bad[:10]


def test_abi_number_of_header_and_host_agree():
    """include/w2s.h's W2S_ABI_VERSION is duplicated by hand in wav2sleep_amd/lib.py (and compiled into the library): the three must move together."""
    from wav2sleep_amd import lib
    m = re.search(r'^#define\s+W2S_ABI_VERSION\s+(\d+)', open(HEADER).read(), re.M)
    assert m, 'W2S_ABI_VERSION not found in include/w2s.h'
    assert int(m.group(1)) == lib.ABI_VERSION
    dll = lib.load()
    assert dll.w2s_abi_version() == lib.ABI_VERSION


def test_no_kernel_spills_beyond_the_allowed_list():
    """Register spills are silent (the units built with `-mllvm -amdgpu-mfma-vgpr-form` hold their accumulators in the 256 architectural
    VGPRs): every kernel's scratch size from the code-object metadata, against wav2sleep_amd/isa_audit.SCRATCH_ALLOWED."""
    from wav2sleep_amd import isa_audit, lib
    if not os.path.exists(isa_audit.READELF):
        pytest.skip('llvm-readelf not in this image')
    res = isa_audit.resources(lib.LIB_PATH)
    assert len(res) > 300, len(res)   # the notes really list the kernels
    hot = [k for k in res if 'bwd_fused_bf_kernel' in k or 'conv_fwd_bf_kernel' in k]
    assert hot and all(res[k]['scratch'] == 0 for k in hot), [(k, res[k]) for k in hot if res[k]['scratch']]
    assert not isa_audit.scratch_violations(lib.LIB_PATH), isa_audit.scratch_violations(lib.LIB_PATH)
