"""What does a plain streaming read reach on this GPU?  torch.sum over a buffer of T's size (1.96 GB fp64) and over 4x that, and a
device-to-device copy, timed with events: the practical ceiling the contraction kernel's 6.2 TB/s should be held against."""
import torch

assert torch.cuda.is_available()
dev = torch.device("cuda:0")
for gb in (1.96, 7.84):
    n = int(gb * 1e9 / 8)
    x = torch.ones(n, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    for name, fn, bytes_ in (("sum (read)", lambda: x.sum(), 8 * n), ("copy (read + write)", lambda: y.copy_(x), 16 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%.2f GB %-20s %.3f ms  %.2f TB/s" % (gb, name, ms, bytes_ / ms / 1e9), flush=True)
    del x, y
