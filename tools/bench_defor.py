#!/usr/bin/env python3
"""Times the deformable FilterInterpolation variants at a padded 1080p frame.
python tools/bench_defor.py [channels] [filter size: 4 or 6]   (staged kernel and, second column, the general gather kernel)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 3
FS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = S.padded_size(1080, 1920)
gen = S.generator()
img = (S.frames(1, H, W, gen) if C == 3 else S.context(1, C, H, W, gen)).cuda()
flow = S.flow(1, H, W, 8.0, gen, "smooth").cuda()
filt = (S.filters(1, H, W, gen) if FS == 4 else torch.rand((1, FS * FS, H, W), generator=gen)).cuda()
off = (torch.randn((1, 2 * FS * FS, H, W), generator=gen) * 0.5).cuda()
out = torch.empty_like(img)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


px = H * W
for variant, name in ((0, "offset (A1b)"), (1, "region (A1c)"), (2, "nofilter (A1d)")):
    third = off if variant == 2 else filt
    ms = timed(lambda: cabi.filterinterp_forward_defor(variant, img, flow, third, off, out))
    mg = timed(lambda: cabi.filterinterp_forward_defor(variant, img, flow, third, off, out, general=True), 5)
    byts = (2 + (2 if variant == 2 else 3) * FS * FS + 2 * C) * 4.0 * px
    print("defor %-15s fs=%d C=%-3d staged %8.4f ms (%7.1f GB/s algorithmic)  general gather %8.4f ms  (%.1f x)"
          % (name, FS, C, ms, byts / ms / 1e6, mg, mg / ms), flush=True)
if FS == 4:
    ms = timed(lambda: cabi.filterinterp_forward_ori(img, flow, filt, out))
    print("ori                   C=%-3d %8.4f ms" % (C, ms))
