#!/usr/bin/env python3
"""Where a channel step of the shared-window kernel spends its time (development build with -DFM_STAMPS):
    make -C <pkg>/csrc OUT=../lib_dev EXTRA="-DVFI_DEV -DFM_STAMPS" && python tools/fm_stamps.py [--kpair K] [--flow smooth] [--nt 3]
Per wave and steady-state step, in s_memtime ticks: staging issue, compute (tap reads + arithmetic + stores), the wait for
the next window (vmcnt), the barrier."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402
vfidkr_amd.LIB_PATH = os.path.join(ROOT, "video-frame-interpolation-based-on-deformable-kernel-region_amd", "lib_dev", "libvfi_hip.so")
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kpair", type=int, default=18)
ap.add_argument("--flow", default="smooth")
ap.add_argument("--nt", type=int, default=3)
args = ap.parse_args()
lib = cabi.lib()
lib.vfi_dev_multi(args.kpair, 3)
dev = torch.device("cuda:0")
h, w = S.padded_size(1080, 1920)
gen = S.generator()
base = S.flow(1, h, w, 8.0, gen, args.flow)
depth = S.depth_weight(1, h, w, gen).to(dev)
ctx = S.context(1, 196, h, w, gen).to(dev)
filt = S.filters(1, h, w, gen).to(dev)
projs = []
for t in (0.25, 0.5, 0.75)[:args.nt]:
    c, o = torch.empty((1, 1, h, w), device=dev), torch.empty((1, 2, h, w), device=dev)
    assert cabi.depthflowprojection_forward((base * (2.0 * t)).contiguous().to(dev), depth, c, o, 1) == 0
    projs.append(o)
outs = [torch.empty_like(ctx) for _ in range(args.nt)]
buf = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    assert cabi.filterinterp_forward_ori_multi(ctx, projs, filt, outs) == 0
lib.vfi_dev_multi_stamps(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    assert cabi.filterinterp_forward_ori_multi(ctx, projs, filt, outs) == 0
e1.record()
torch.cuda.synchronize()
lib.vfi_dev_multi_stamps(buf)
v = list(buf)
steps = max(1, v[4])
tot = sum(v[:4])
print("kpair %d %s nt %d: %.1f us per launch; per wave-step (ticks): issue %.0f  compute %.0f  vmcnt wait %.0f  barrier %.0f  = %.0f  (%d wave-steps per launch)"
      % (args.kpair, args.flow, args.nt, e0.elapsed_time(e1) / n * 1e3, v[0] / steps, v[1] / steps, v[2] / steps, v[3] / steps, tot / steps, steps // n))
