#!/usr/bin/env python3
"""Random-shape soak of the PWC-configuration correlation kernels: the aligned path (corr_forward_k1_quad / _rows2, chosen by size)
against the same call on inputs that sit at an odd element offset of their storage (rows not 16-byte aligned: the one-pixel
tiled kernels / the one-thread-per-output kernel) -- every kernel sums the channels in the same order, so the bits must agree.
    python tools/corr_soak.py [--cases 60] [--seed 1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
g = torch.Generator().manual_seed(args.seed)
dev = torch.device("cuda:0")
bad = 0
for case in range(args.cases):
    b = int(torch.randint(1, 3, (1,), generator=g))
    c = int(torch.randint(1, 70, (1,), generator=g))
    h = int(torch.randint(8, 300, (1,), generator=g))
    w = 4 * int(torch.randint(4, 260, (1,), generator=g))
    pad = 4 if case % 5 else 0
    if pad == 0 and (h <= 8 or w <= 8):
        continue
    n = b * c * h * w
    s1, s2 = torch.randn(n + 1, generator=g).to(dev), torch.randn(n + 1, generator=g).to(dev)
    a1, a2 = s1[:n].view(b, c, h, w), s2[:n].view(b, c, h, w)                       # aligned
    u1, u2 = s1[1:].view(b, c, h, w), s2[1:].view(b, c, h, w)                       # 4 bytes off
    u1.copy_(a1.clone()); u2.copy_(a2.clone())
    a1, a2 = u1.clone(), u2.clone()                                                 # fresh, aligned allocations of the same values
    ref = cabi.correlation_forward(u1, u2, pad, 1, 4, 1, 1)
    out = cabi.correlation_forward(a1, a2, pad, 1, 4, 1, 1)
    pa, pb = cabi.correlation_forward_pair(a1, a2, a2, a1, pad, 1, 4, 1, 1)
    ok = torch.equal(out.view(torch.int32), ref.view(torch.int32)) and torch.equal(pa, out)
    ok = ok and torch.equal(pb, cabi.correlation_forward(a2, a1, pad, 1, 4, 1, 1))
    if not ok:
        bad += 1
        print("MISMATCH", (b, c, h, w, pad), flush=True)
    if (case + 1) % 20 == 0:
        print("%d cases, %d with mismatches" % (case + 1, bad), flush=True)
print("done: %d cases, %d with mismatches" % (args.cases, bad))
sys.exit(1 if bad else 0)
