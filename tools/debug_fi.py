#!/usr/bin/env python3
"""Where does the LDS FilterInterpolation path differ from the direct kernel?  python tools/debug_fi.py B C H W"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi  # noqa: E402

B, C, H, W = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (1, 3, 32, 48)
g = torch.Generator().manual_seed(1)
img = torch.rand((B, C, H, W), generator=g).cuda()
filt = torch.rand((B, 16, H, W), generator=g).cuda()
flow = (torch.rand((B, 2, H, W), generator=g) * 4 - 2).cuda()
a = torch.full_like(img, float("nan"))
b = torch.full_like(img, float("nan"))
assert cabi.filterinterp_forward_ori(img, flow, filt, a) == 0
assert cabi.filterinterp_forward_ori(img, flow, filt, b, direct=True) == 0
a, b = a.cpu().numpy(), b.cpu().numpy()
bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
print("max abs diff", np.nanmax(np.abs(a - b)), "max ulp-ish", np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-30)))
print("mismatches", bad.sum(), "nan in lds", np.isnan(a).sum(), "nan in direct", np.isnan(b).sum())
for bb in range(B):
    for c in range(C):
        m = bad[bb, c]
        if m.any():
            ys, xs = np.nonzero(m)
            print("b %d c %d: %d bad, y %d..%d x %d..%d, nan %d; first (y,x,lds,direct): " % (
                bb, c, m.sum(), ys.min(), ys.max(), xs.min(), xs.max(), np.isnan(a[bb, c]).sum()),
                [(int(y), int(x), float(a[bb, c, y, x]), float(b[bb, c, y, x])) for y, x in list(zip(ys, xs))[:4]])
