#!/usr/bin/env python3
"""fp16-storage direct kernel vs the fp32 direct kernel on the widened image, rounded once."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vfidkr_amd  # noqa
from vfidkr_amd import cabi
B, C, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (2, 5, 40, 72)
g = torch.Generator().manual_seed(3)
img16 = torch.rand((B, C, H, W), generator=g).to(torch.float16).cuda()
filt = (torch.rand((B, 16, H, W), generator=g) * 0.25).cuda()
flow = (torch.rand((B, 2, H, W), generator=g) * 6 - 3).cuda()
o16 = torch.empty_like(img16)
assert cabi.filterinterp_forward_ori_f16(img16, flow, filt, o16, direct=True) == 0
o32 = torch.empty((B, C, H, W), device="cuda")
assert cabi.filterinterp_forward_ori(img16.float(), flow, filt, o32, direct=True) == 0
ref = o32.to(torch.float16)
bad = (ref != o16)
print("mismatches", int(bad.sum()), "of", bad.numel())
idx = bad.nonzero()[:6]
for i in idx:
    b, c, y, x = (int(v) for v in i)
    print(b, c, y, x, "f16 kernel", float(o16[b, c, y, x]), "fp32->half", float(ref[b, c, y, x]), "fp32", float(o32[b, c, y, x]),
          "flow", float(flow[b, 0, y, x]), float(flow[b, 1, y, x]))
