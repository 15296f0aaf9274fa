#!/usr/bin/env python3
"""Runs a few shared-window FilterInterpolation launches (fi_forward_ori_multi<NT>, C=196, 1152x1984) for rocprofv3 passes.
    rocprofv3 --pmc ... --kernel-trace --output-format csv -d out -- python3 tools/prof_multi.py [flow] [nt] [--lib L] [--kpair K]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
kpair = None
if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
if "--kpair" in sys.argv:
    i = sys.argv.index("--kpair")
    kpair = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "smooth"
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if kpair is not None:
    cabi.lib().vfi_dev_multi(kpair, 3)
dev = torch.device("cuda:0")
h, w = S.padded_size(1080, 1920)
gen = S.generator()
base = S.flow(1, h, w, 8.0, gen, model)
depth = S.depth_weight(1, h, w, gen).to(dev)
ctx = S.context(1, 196, h, w, gen).to(dev)
filt = S.filters(1, h, w, gen).to(dev)
projs = []
for t in (0.25, 0.5, 0.75)[:nt]:
    c, o = torch.empty((1, 1, h, w), device=dev), torch.empty((1, 2, h, w), device=dev)
    assert cabi.depthflowprojection_forward((base * (2.0 * t)).contiguous().to(dev), depth, c, o, 1) == 0
    projs.append(o)
outs = [torch.empty_like(ctx) for _ in range(nt)]
for _ in range(4):
    assert cabi.filterinterp_forward_ori_multi(ctx, projs, filt, outs) == 0
torch.cuda.synchronize()
