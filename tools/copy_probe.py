"""Dev tool (GPU box): what a plain copy of the blur's bytes achieves on this box -- torch.copy_ of the pitched [8,64,513(516),513] input
cropped to [8,64,512,512] (the blur's access pattern without the arithmetic) and of a dense tensor of the same total size."""
import torch

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

buf = torch.randn(8, 64, 513, 516, device='cuda')
dst = torch.empty(8, 64, 512, 512, device='cuda')
nbytes = 4 * (8 * 64 * 513 * 513 + 8 * 64 * 512 * 512)            # what the blur moves
t = timeit(lambda: dst.copy_(buf[:, :, :512, :512]))
print(f'copy_ pitched crop [8,64,512(516),512] -> dense: {t*1e6:.1f} us  {2 * dst.numel() * 4 / t / 1e9:.0f} GB/s on its own bytes ({nbytes / t / 1e9:.0f} GB/s if priced as the blur)')
a = torch.randn(nbytes // 8, device='cuda'); b = torch.empty_like(a)
t = timeit(lambda: b.copy_(a))
print(f'copy_ dense {a.numel() * 4 / 1e6:.0f} MB -> {a.numel() * 4 / 1e6:.0f} MB: {t*1e6:.1f} us  {nbytes / t / 1e9:.0f} GB/s')
