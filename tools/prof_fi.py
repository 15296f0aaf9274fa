#!/usr/bin/env python3
"""Runs a few FilterInterpolation launches for rocprofv3 counter passes.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/prof_fi.py [flow] [C]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
if "--lib" in sys.argv:         # a development build of the library (make OUT=../lib_dev EXTRA=-DVFI_DEV) for the knobs
    i = sys.argv.index("--lib")
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "smooth"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 196
if len(sys.argv) > 3:        # development knobs of the LDS kernel (flags[:groups])
    fl = sys.argv[3].split(":")
    cabi.lib().vfi_dev_filterinterp(int(fl[0], 0), int(fl[1], 0) if len(fl) > 1 else 0)
dev = torch.device("cuda:0")
h, w = S.padded_size(1080, 1920)
gen = S.generator()
img = (S.context(1, C, h, w, gen) if C != 3 else S.frames(1, h, w, gen)).to(dev)
filt = S.filters(1, h, w, gen).to(dev)
if model == "invalid":       # |fx| >= W/2 everywhere: pure copy-through, known traffic (C planes in + out, flow)
    flow = torch.full((1, 2, h, w), float(w), device=dev)
elif model == "zero":        # windows = tile + 3: known staging pattern
    flow = torch.zeros((1, 2, h, w), device=dev)
else:
    flow = S.flow(1, h, w, 8.0, gen, model).to(dev)
out = torch.empty_like(img)
for _ in range(4):
    assert cabi.filterinterp_forward_ori(img, flow, filt, out) == 0
torch.cuda.synchronize()
count = torch.zeros((1, 1, h, w), device=dev)
proj = torch.zeros((1, 2, h, w), device=dev)
for _ in range(3):
    count.zero_(); proj.zero_()
    assert cabi.flowprojection_forward(flow, count, proj, 1) == 0
torch.cuda.synchronize()
