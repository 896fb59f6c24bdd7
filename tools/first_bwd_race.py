"""enc_first_bwd_kernel in isolation: same inputs, repeated launches, with and without another stream keeping the GPU busy (a device copy,
a second instance, the split-precision fused backward / wide conv kernels looping -- what shares the CUs with it inside a train step).
W2S_LIB=build_alt/libw2s_f4.so (tools/altlib.sh f4 "-DW2S_FIRST_BWD_FLOAT4=1" enc_misc.hip) runs the round-1 float4 form of the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2sleep_amd import lib
os.environ.setdefault('BF', '1')
from tools import kbench
dev = torch.device('cuda')
torch.manual_seed(0)
B, L = 2, 983040
x = torch.randn(B, L, device=dev); gn1 = torch.randn(B, L, 16, device=dev) * 0.01; gpre = torch.randn(B, L // 2, 16, device=dev) * 0.01
st1 = torch.stack([torch.randn(B, 16, device=dev) * 0.1, torch.rand(B, 16, device=dev) + 0.5], -1).contiguous()
bs1 = (torch.randn(B, 16, 2, device=dev) * 1e-3).contiguous()
w1 = torch.randn(16, 1, 3, device=dev) * 0.5
nslab = 480
def run():
    slab = torch.full((nslab, 64), float('nan'), device=dev)
    lib.enc_first_bwd(x, gn1, None, st1, bs1, gpre, slab, nslab, B, L, 16, w1=w1, causal=False)
    return slab
ref = run(); torch.cuda.synchronize()
hogs = {'with bwd_fused 16ch on a second stream': kbench.CASES['b16']()[0], 'with bwd_fused 32ch on a second stream': kbench.CASES['b32']()[0],
        'with conv_wide 64ch dgrad on a second stream': kbench.CASES['d64']()[0]}
only = os.environ.get('MODES')   # comma-separated substrings: run only the matching modes
for mode in ('alone', 'with a copy stream', 'two instances on two streams', *hogs):
    if only and not any(o in mode for o in only.split(',')):
        continue
    side = torch.cuda.Stream()
    big_a = torch.randn(1 << 28, device=dev); big_b = torch.empty_like(big_a)
    bad = 0
    for it in range(int(os.environ.get('RUNS', 100))):
        if mode in hogs:
            with torch.cuda.stream(side):
                for _ in range(3): hogs[mode]()
        if mode == 'with a copy stream':
            with torch.cuda.stream(side):
                for _ in range(3): big_b.copy_(big_a)
        if mode == 'two instances on two streams':
            with torch.cuda.stream(side):
                s2 = run()
        s = run()
        torch.cuda.synchronize()
        bad += int(not torch.equal(s, ref))
        if mode == 'two instances on two streams':
            bad += int(not torch.equal(s2, ref))
    print(f'{mode:46s}: {bad} launches differ from the first result; columns that differ in the last: {sorted(set((s != ref).nonzero()[:, 1].tolist()))[:20]}')
