#!/usr/bin/env python3
"""A few MinDepthFlowProjection forwards at padded 1080p, for rocprofv3 --kernel-trace --stats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

H, W = S.padded_size(1080, 1920)
gen = S.generator()
flow = S.flow(1, H, W, 8.0, gen, "smooth").cuda()
wd = (torch.rand((1, 1, H, W), generator=gen) + 0.1).cuda()
c2, o2 = torch.zeros((1, 1, H, W), device="cuda"), torch.zeros_like(flow)
for _ in range(10):
    c2.zero_(), o2.zero_()
    assert cabi.mindepthflowprojection_forward(flow, wd, c2, o2, 1) == 0
torch.cuda.synchronize()
print("holes before fill: %.3f" % float((c2 <= 0).float().mean()))
