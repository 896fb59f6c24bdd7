"""Post-build audit of libw2s_hip.so's gfx950 code for an instruction form that returns wrong results on MI355X.

Observed on gfx950 (tools/pk_fma_opsel_repro.hip): a packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32) whose LOW lane reads the HIGH half of its SECOND source (`op_sel:[x,1,...]`) returns a wrong low-lane result, a few per cent
of the time, while another wave on the same CU executes v_mfma_f32_16x16x32_bf16 -- which is what every split-precision kernel of this
library does on the neighbouring streams.  hipcc emits that form where it vectorises scalar code or lowers a shuffle; nothing in the
source says so.  This walks every gfx950 code object bundled in the shared library, disassembles it with llvm-objdump and reports every
such instruction with its kernel.  `lib.build()` and tests/test_cabi_cpu.py run it: a build that contains one fails.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import tempfile

import shutil

_MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
_CCOB = b'CCOB'   # header of a COMPRESSED offload bundle (hipcc --offload-compress): not readable here, must not pass as "nothing found"


def _find_objdump() -> str:
    """llvm-objdump of the ROCm installation that built the library: $ROCM_PATH, next to hipcc, then the image's default."""
    cands = []
    if os.environ.get('ROCM_PATH'):
        cands.append(os.path.join(os.environ['ROCM_PATH'], 'lib', 'llvm', 'bin', 'llvm-objdump'))
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), 'lib', 'llvm', 'bin', 'llvm-objdump'))
    cands.append('/opt/rocm/lib/llvm/bin/llvm-objdump')
    for c in cands:
        if os.path.exists(c):
            return c
    return shutil.which('llvm-objdump') or cands[-1]


OBJDUMP = _find_objdump()
MIN_PACKED = 100000   # the shipped library holds ~4e5 packed-fp32 instructions: far fewer means the audit did not see the code
_BAD = re.compile(r'\bv_pk_(?:fma|mul|add)_f32\b.*\bop_sel:\[[01],1')


def code_objects(path: str):
    """(target triple, ELF bytes) of every gfx950 code object in the clang offload bundles of a HIP shared library / object file."""
    data = open(path, 'rb').read()
    pos = 0
    while True:
        i = data.find(_MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from('<Q', data, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', data, p)
            p += 24
            triple = data[p:p + tl].decode(errors='replace')
            p += tl
            if 'gfx950' in triple and size:
                yield triple, data[i + off:i + off + size]
        pos = i + len(_MAGIC)


def audit(path: str) -> tuple[int, list[tuple[str, str]]]:
    """-> (number of packed-fp32 instructions seen, [(kernel symbol, instruction text)] of the vulnerable ones)."""
    if not os.path.exists(OBJDUMP):
        raise RuntimeError(f'{OBJDUMP} not found: cannot audit {path}')
    seen, bad = 0, []
    blobs = list(code_objects(path))
    if not blobs:
        raw = open(path, 'rb').read()
        raise RuntimeError(f'{path}: no gfx950 code object found in an uncompressed clang offload bundle'
                           + (' (the file holds COMPRESSED bundles: build without --offload-compress)' if _CCOB in raw else '')
                           + ' -- the audit cannot vouch for this build')
    for _, blob in blobs:
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(blob)
        try:
            out = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        sym = '?'
        for ln in out.splitlines():
            m = re.match(r'^[0-9a-f]+ <(\S+)>:', ln)
            if m:
                sym = m.group(1)
            elif 'v_pk_' in ln and '_f32' in ln:
                seen += 1
                if _BAD.search(ln):
                    bad.append((sym, ln.split('//')[0].strip()))
    return seen, bad


READELF = os.path.join(os.path.dirname(OBJDUMP), 'llvm-readelf')
# Kernels that are ALLOWED scratch memory (bytes per lane), by mangled-name prefix.  Everything else must have none: several translation units
# are built with `-mllvm -amdgpu-mfma-vgpr-form` (csrc/Makefile), which moves MFMA accumulators into the 256 architectural VGPRs -- a kernel
# that outgrows them then spills silently.  The entries: two tile configurations of the generic conv that `pick_cfg` never selects (NT = 8 with
# WN = 2; dead instantiations), the exact-fp32 16-channel fused backward (W2S_EXACT_FP32=1 only) and the 128-channel transposed data gradient
# (12 B, measured round 4: the non-spilling tile was slower).
SCRATCH_ALLOWED = {
    '_Z14conv_cl_kernelILi8ELi4ELi3ELi3ELi1ELi2E': 320, '_Z14conv_cl_kernelILi8ELi4ELi4ELi4ELi1ELi2E': 320, '_Z14conv_cl_kernelILi8ELi4ELi7ELi1ELi0ELi2E': 320,
    '_Z14conv_cl_kernelILi8ELi4ELi7ELi1ELi1ELi2E': 320, '_Z16bwd_fused_kernelILi1ELi1ELi4ELi0ELi1EE': 32, '_Z16conv_wide_kernelILi8ELi8ELi4ELi1ELi5ELi4ELi4ELi1ELi1E': 16,
}


def resources(path: str) -> dict[str, dict[str, int]]:
    """kernel symbol -> {vgpr, agpr, sgpr, scratch (bytes per lane), spill (VGPRs), lds} from the code objects' metadata notes."""
    if not os.path.exists(READELF):
        raise RuntimeError(f'{READELF} not found: cannot read kernel resources of {path}')
    out = {}
    for _, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.co', delete=False) as f:
            f.write(blob)
        try:
            txt = subprocess.run([READELF, '--notes', f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        for blk in txt.split('  - .agpr_count')[1:]:
            blk = '.agpr_count' + blk

            def num(key, blk=blk):
                m = re.search(r'\.' + key + r':\s+(\d+)', blk)
                return int(m.group(1)) if m else 0
            m = re.search(r'^\s+\.name:\s+(\S+)', blk, re.M)
            if m:
                out[m.group(1)] = dict(vgpr=num('vgpr_count'), agpr=num('agpr_count'), sgpr=num('sgpr_count'), scratch=num('private_segment_fixed_size'),
                                       spill=num('vgpr_spill_count'), lds=num('group_segment_fixed_size'))
    return out


def scratch_violations(path: str) -> list[tuple[str, int]]:
    """[(kernel, scratch bytes)] of every kernel that uses scratch memory beyond SCRATCH_ALLOWED."""
    bad = []
    for k, r in resources(path).items():
        lim = max([v for pfx, v in SCRATCH_ALLOWED.items() if k.startswith(pfx)], default=0)
        if r['scratch'] > lim:
            bad.append((k, r['scratch']))
    return bad


if __name__ == '__main__':
    import sys
    from wav2sleep_amd.lib import LIB_PATH
    n, bad = audit(sys.argv[1] if len(sys.argv) > 1 else LIB_PATH)
    print(f'{n} packed-fp32 instructions, {len(bad)} with the low lane reading the high half of src1')
    for sym, ins in bad:
        print(f'  {sym}: {ins}')
    sv = scratch_violations(sys.argv[1] if len(sys.argv) > 1 else LIB_PATH)
    print(f'{len(sv)} kernels with scratch memory beyond the allowed list')
    for k, b in sv:
        print(f'  {k}: {b} B per lane')
    sys.exit(1 if (bad or sv) else 0)
