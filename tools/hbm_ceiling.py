#!/usr/bin/env python3
"""HBM ceiling on this box for the FilterInterpolation C=196 footprint: time plain device copies of the
same planes (read N bytes + write N bytes) so roofline fractions can be read against what a copy achieves."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


dev = torch.device("cuda:0")
for C in (196, 64, 3):
    x = torch.randn((1, C, 1152, 1984), device=dev)
    y = torch.empty_like(x)
    nbytes = 2 * x.numel() * 4
    ms = timed(lambda: y.copy_(x))
    print("copy_      C=%3d %8.4f ms %7.1f GB/s (read+write)" % (C, ms, nbytes / ms / 1e6))
    ms = timed(lambda: torch.add(x, 1.0, out=y))
    print("add        C=%3d %8.4f ms %7.1f GB/s (read+write)" % (C, ms, nbytes / ms / 1e6))
    ms = timed(lambda: y.fill_(1.0))
    print("fill_      C=%3d %8.4f ms %7.1f GB/s (write only)" % (C, ms, nbytes / 2 / ms / 1e6))
    ms = timed(lambda: x.sum())
    print("sum        C=%3d %8.4f ms %7.1f GB/s (read only)" % (C, ms, nbytes / 2 / ms / 1e6))
