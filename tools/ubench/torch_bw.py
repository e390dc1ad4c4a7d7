#!/usr/bin/env python3
"""Ceiling check for write-heavy kernels: fill (write only), copy (1 read : 1 write) and a 1 read : 2 write split over 1 GiB buffers (torch kernels, HIP events)."""
import torch


def t(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


n = 256 << 20
x = torch.rand(n, device="cuda"); y = torch.empty_like(x); z = torch.empty(2 * n, device="cuda")
for _ in range(3):
    s = t(lambda: y.fill_(1.0)); print(f"fill   {4 * n / s / 1e12:.2f} TB/s written")
    s = t(lambda: y.copy_(x)); print(f"copy   {8 * n / s / 1e12:.2f} TB/s moved (1 read : 1 write)")
    s = t(lambda: torch.stack((x, x), dim=1, out=z.view(n, 2))); print(f"1r:2w  {12 * n / s / 1e12:.2f} TB/s moved")
