#!/usr/bin/env python3
"""FilterInterpolation C=196 at padded 1080p against the flow magnitude (smooth model, sigma in pixels): how the
launch time follows the size of the staged windows.  python tools/fi_sigma.py [sigmas]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

sigmas = [float(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0.0, 2.0, 4.0, 8.0, 12.0, 16.0, 24.0]
H, W = S.padded_size(1080, 1920)
gen = S.generator()
ctx = S.context(1, 196, H, W, gen).cuda()
filt = S.filters(1, H, W, gen).cuda()
out = torch.empty_like(ctx)
for sg in sigmas:
    flow = S.flow(1, H, W, sg, gen, "smooth").cuda()
    for _ in range(3):
        cabi.filterinterp_forward_ori(ctx, flow, filt, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        cabi.filterinterp_forward_ori(ctx, flow, filt, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("sigma %5.1f px   %8.4f ms   %7.1f GB/s algorithmic" % (sg, ms, 1640.0 * H * W / ms / 1e6), flush=True)
