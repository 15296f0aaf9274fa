#!/usr/bin/env python3
"""Parity of the LDS FilterInterpolation kernel under a development flag word against the direct kernel.
python tools/debug_fi_flags.py FLAGS [B C H W sigma]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

flags = int(sys.argv[1], 0)
shapes = [(1, 5, 70, 200, 3.0), (2, 3, 33, 65, 1.0), (1, 7, 128, 256, 8.0), (1, 4, 64, 128, 40.0)]
if len(sys.argv) > 6:
    shapes = [tuple(int(v) for v in sys.argv[2:6]) + (float(sys.argv[6]),)]
gen = S.generator()
ok = True
for (B, C, H, W, sig) in shapes:
    img = torch.randn((B, C, H, W), generator=gen).cuda()
    filt = S.filters(B, H, W, gen).cuda()
    flow = S.flow(B, H, W, sig, gen, "smooth").cuda()
    a = torch.full_like(img, float("nan"))
    b = torch.full_like(img, float("nan"))
    cabi.lib().vfi_debug_filterinterp(ctypes.c_int(flags), ctypes.c_int(0))
    assert cabi.filterinterp_forward_ori(img, flow, filt, a) == 0
    cabi.lib().vfi_debug_filterinterp(ctypes.c_int(0), ctypes.c_int(0))
    assert cabi.filterinterp_forward_ori(img, flow, filt, b, direct=True) == 0
    same = torch.equal(a, b)
    ok &= same
    print((B, C, H, W, sig), "flags 0x%x" % flags, "bit-identical" if same else "MISMATCH %d" % int((a != b).sum()))
sys.exit(0 if ok else 1)
