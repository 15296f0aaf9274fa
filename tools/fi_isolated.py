#!/usr/bin/env python3
"""Development: the C=196 FilterInterpolation launch timed in isolation (a short kernel in front, a synchronise behind,
as inside bench.py's step) for several uniform channel-group splits (vfi_dev_filterinterp groups knob, -DVFI_DEV build).
Back-to-back loops hide the tail of a launch under the head of the next one; this does not.
    python tools/fi_isolated.py          (needs <pkg>/lib_dev)
"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import vfidkr_amd
vfidkr_amd.LIB_PATH = os.path.join(ROOT, "video-frame-interpolation-based-on-deformable-kernel-region_amd", os.environ.get("FI_LIBDIR", "lib_dev"), "libvfi_hip.so")
from vfidkr_amd import cabi, synthetic as S
dev = torch.device("cuda:0")
h, w = S.padded_size(1080, 1920)
gen = S.generator()
ctx = [S.context(1, 196, h, w, gen).to(dev) for _ in range(2)]
filt = S.filters(1, h, w, gen).to(dev)
flow = S.flow(1, h, w, 8.0 * w / 1984.0, gen, "smooth").to(dev)
out = torch.empty_like(ctx[0])
depth = S.depth_weight(1, h, w, gen).to(dev)
count = torch.zeros((1, 1, h, w), device=dev); proj = torch.zeros((1, 2, h, w), device=dev)
knob = cabi.lib().vfi_dev_filterinterp
def iso(groups, n=40, flags=8):
    knob(flags, groups)
    ts = []
    for i in range(n + 5):
        cabi.depthflowprojection_forward(flow, depth, count, proj, 1)          # something short in front, as in the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); cabi.filterinterp_forward_ori(ctx[i % 2], flow, filt, out); e1.record()
        torch.cuda.synchronize()
        if i >= 5: ts.append(e0.elapsed_time(e1))
    ts.sort()
    return sum(ts) / len(ts), ts[len(ts) // 2], ts[0]
import sys as _sys
FLAGS = [int(v, 0) for v in _sys.argv[1].split(",")] if len(_sys.argv) > 1 else [8]
GROUPS = [int(v) for v in _sys.argv[2].split(",")] if len(_sys.argv) > 2 else [1, 2, 3, 4, 6, 8]
for rep in range(2):
    for fl in FLAGS:
        for g in GROUPS:
            print("flags %#x groups knob %d: mean %.4f median %.4f min %.4f ms" % ((fl, g) + iso(g, flags=fl)), flush=True)
