#!/usr/bin/env python3
"""What this box's memory system sustains for plain streaming fills / copies / reads (torch kernels), to price the GEMM epilogues against."""
import time
import torch
n = 1 << 28   # 1 GiB of f32
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, it=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it
gb = n * 4 / 1e9
print("fill   %.2f TB/s written" % (gb / t(lambda: a.fill_(1.0)) / 1e3))
print("copy   %.2f TB/s read + %.2f TB/s written" % ((gb / t(lambda: b.copy_(a)) / 1e3,) * 2))
print("sum    %.2f TB/s read" % (gb / t(lambda: a.sum()) / 1e3))
h = torch.empty(n // 2, dtype=torch.float16, device="cuda")
print("f32->f16 convert  %.2f TB/s read + %.2f TB/s written" % (gb / t(lambda: h.copy_(a[: n // 2])) / 2e3, gb / t(lambda: h.copy_(a[: n // 2])) / 4e3))
