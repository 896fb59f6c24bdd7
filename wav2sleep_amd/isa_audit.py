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

_MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
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
    for _, blob in code_objects(path):
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


if __name__ == '__main__':
    import sys
    from wav2sleep_amd.lib import LIB_PATH
    n, bad = audit(sys.argv[1] if len(sys.argv) > 1 else LIB_PATH)
    print(f'{n} packed-fp32 instructions, {len(bad)} with the low lane reading the high half of src1')
    for sym, ins in bad:
        print(f'  {sym}: {ins}')
    sys.exit(1 if bad else 0)
