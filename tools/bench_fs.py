#!/usr/bin/env python3
"""FilterInterpolation (_ori) forward at padded 1080p for the reference's filter sizes: LDS-staged vs direct kernel.
python tools/bench_fs.py [channels]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W = S.padded_size(1080, 1920)
gen = S.generator()
img = (S.frames(1, H, W, gen) if C == 3 else S.context(1, C, H, W, gen)).cuda()
flow = S.flow(1, H, W, 8.0, gen, "smooth").cuda()
out = torch.empty_like(img)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for fs in (2, 4, 5, 6):
    filt = torch.rand((1, fs * fs, H, W), generator=gen).cuda()
    a = timed(lambda: cabi.filterinterp_forward_ori(img, flow, filt, out))
    b = timed(lambda: cabi.filterinterp_forward_ori(img, flow, filt, out, direct=True), 3)
    byts = (2 + fs * fs + 2 * C) * 4.0 * H * W
    print("fs=%d C=%-3d  LDS %8.4f ms (%6.1f GB/s algorithmic)   direct %8.4f ms" % (fs, C, a, byts / a / 1e6, b), flush=True)
