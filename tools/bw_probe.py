"""HBM streaming ceilings on this box with plain torch ops (copy = 1:1 read/write,
sum = read only, fill = write only), footprints well past the 256 MB Infinity Cache."""
import torch, time
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (128, 512, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, dtype=torch.int32, device=dev).fill_(1)
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a)); print("copy  %5d MB each: %.2f TB/s (r+w)" % (mb, 2 * n * 4 / t / 1e12))
    t = timeit(lambda: a.sum());    print("sum   %5d MB     : %.2f TB/s (r)" % (mb, n * 4 / t / 1e12))
    t = timeit(lambda: b.fill_(3)); print("fill  %5d MB     : %.2f TB/s (w)" % (mb, n * 4 / t / 1e12))
    c = torch.empty_like(a)
    t = timeit(lambda: torch.add(a, b, out=c)); print("add   %5d MB each: %.2f TB/s (2r+w)" % (mb, 3 * n * 4 / t / 1e12))
    del a, b, c
