"""HBM read / write / copy rates with stock torch kernels on 1 GiB buffers (reference points for the roofline)."""
import torch, time
n = 256 * 1024 * 1024
a = torch.randn(n, device='cuda'); b = torch.empty_like(a)
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for name, fn, byt in (('fill (write only)', lambda: b.fill_(1.0), 4 * n), ('sum (read only)', lambda: a.sum(), 4 * n), ('copy (read+write)', lambda: b.copy_(a), 8 * n),
                      ('mul (read+write)', lambda: torch.mul(a, 2.0, out=b), 8 * n), ('add3 (2 reads + write)', lambda: torch.add(a, b, out=b), 12 * n)):
    ms = t(fn); print(f'{name:24s} {ms:7.3f} ms  {byt / ms / 1e9:7.2f} TB/s')
