#!/usr/bin/env python3
"""A few launches of one secondary kernel for rocprofv3 counter passes (tools/collect_extras.sh):
    python3 tools/prof_misc.py f16 [flow]   (f32: the fp32 kernel)   FilterInterpolation C=196 with fp16 storage (BASELINE configs[2]), smooth flow
    python3 tools/prof_misc.py corr    correlation at the finest PWC-Net level of a 1080p pair (32 x 288 x 496)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
if "--lib" in sys.argv:         # a development build of the library (tools/mkvariant.sh)
    i = sys.argv.index("--lib")
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

what = sys.argv[1]
dev = torch.device("cuda:0")
gen = S.generator()
h, w = S.padded_size(1080, 1920)
if what in ("f16", "f32"):
    ctx = S.context(1, 196, h, w, gen).to(dev)
    if what == "f16":
        ctx = ctx.half()
    filt = S.filters(1, h, w, gen).to(dev)
    model = sys.argv[2] if len(sys.argv) > 2 else "smooth"
    xs = torch.arange(w, device=dev, dtype=torch.float32).view(1, 1, 1, w).expand(1, 1, h, w)
    zero = torch.zeros((1, 1, h, w), device=dev)
    if model == "zero":
        flow = torch.cat([zero, zero], 1)
    elif model == "shift":          # every pixel the same odd whole-pixel shift
        flow = torch.cat([zero + 3.0, zero + 3.0], 1)
    elif model == "stretch":        # fx grows along x: a lane in twenty skips a column
        flow = torch.cat([0.05 * (xs % 256.0), zero], 1)
    elif model == "compress":       # fx shrinks along x: a lane in twenty repeats a column
        flow = torch.cat([-0.05 * (xs % 256.0) + 13.0, zero], 1)
    elif model == "rowstep":        # fy grows along x: a wave's lanes sit on two or three window rows
        flow = torch.cat([zero, 0.05 * (xs % 256.0)], 1)
    else:
        flow = S.flow(1, h, w, 8.0, gen, model).to(dev)
    flow = flow.contiguous()
    out = torch.empty_like(ctx)
    for _ in range(4):
        assert (cabi.filterinterp_forward_ori_f16 if what == "f16" else cabi.filterinterp_forward_ori)(ctx, flow, filt, out) == 0
elif what == "corr":
    a = torch.randn((1, 32, h // 4, w // 4), generator=gen).to(dev)
    b = torch.randn((1, 32, h // 4, w // 4), generator=gen).to(dev)
    for _ in range(4):
        cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
else:
    raise SystemExit("f16 | corr")
torch.cuda.synchronize()
