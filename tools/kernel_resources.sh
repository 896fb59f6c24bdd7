#!/bin/bash
# usage: tools/kernel_resources.sh file.hip [extra hipcc flags]  -> per kernel: VGPRs, scratch bytes per lane (register spills), occupancy
# (compile-time: hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed)
cd "$(dirname "$0")/../wav2sleep_amd/csrc"
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/res.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'n':m.group(1)}; continue
    if cur is None: continue
    for k,pat in (('v',r'VGPRs: (\d+)'),('s',r'ScratchSize \[bytes/lane\]: (\d+)'),('o',r'Occupancy \[waves/SIMD\]: (\d+)')):
        m=re.search(pat,l)
        if m: cur[k]=m.group(1)
    if 'LDS Size' in l:
        n=subprocess.run(['c++filt',cur['n']],capture_output=True,text=True).stdout.strip().split('(')[0]
        print(f\"{n:72s} VGPR {cur.get('v','?'):>4} scratch {cur.get('s','?'):>4} occ {cur.get('o','?')}\")
        cur=None
"
