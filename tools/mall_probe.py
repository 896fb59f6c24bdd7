"""Does a tensor that was just WRITTEN come back from the 256 MB Infinity Cache when the next kernel reads it?  (MI355X box)
For sizes 32 MB .. 1 GB: time of dst.copy_(mid) right after mid.copy_(src) ("hot": mid was just written) against the same copy after
a 2 GB sweep through other memory ("cold")."""
import torch, time
dev = torch.device('cuda')
flush_a = torch.empty(512 << 20 >> 2, device=dev); flush_b = torch.empty_like(flush_a)
def t_copy(dst, src, n=1):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): dst.copy_(src)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for mb in (32, 64, 96, 128, 192, 256, 512, 1024):
    n = (mb << 20) >> 2
    src = torch.randn(n, device=dev); mid = torch.empty_like(src); dst = torch.empty_like(src)
    hot, cold = [], []
    for rep in range(5):
        flush_b.copy_(flush_a); flush_a.copy_(flush_b)          # evict
        mid.copy_(src)                                            # producer: writes mid
        hot.append(t_copy(dst, mid))                              # consumer right behind it
        flush_b.copy_(flush_a); flush_a.copy_(flush_b)
        cold.append(t_copy(dst, mid))
    h, c = sorted(hot)[2], sorted(cold)[2]
    print(f'{mb:5d} MB: consumer hot {h:7.1f} us ({2 * mb / 1024 / h * 1e6 / 1e3:5.2f} TB/s)   cold {c:7.1f} us ({2 * mb / 1024 / c * 1e6 / 1e3:5.2f} TB/s)   hot/cold {h / c:.2f}', flush=True)
