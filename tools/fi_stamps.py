#!/usr/bin/env python3
"""Where a channel step of the single-flow FilterInterpolation kernel spends its time (a build with -DFI_STAMPS):
    tools/mkvariant.sh stamps filterinterp_lds.hip -DFI_STAMPS && python tools/fi_stamps.py --lib <pkg>/lib_vstamps/libvfi_hip.so
Per wave and steady-state step of the skewed 16-byte-staging loop, in s_memtime ticks: staging issue, compute (tap reads +
arithmetic + stores), the wait for the next window (vmcnt), the barrier."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402
if "--lib" in sys.argv:
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--flow", default="smooth")
args = ap.parse_args()
lib = cabi.lib()
dev = torch.device("cuda:0")
h, w = S.padded_size(1080, 1920)
gen = S.generator()
flow = S.flow(1, h, w, 8.0 * w / 1984.0, gen, args.flow).to(dev)
ctx = S.context(1, 196, h, w, gen).to(dev)
filt = S.filters(1, h, w, gen).to(dev)
out = torch.empty_like(ctx)
buf = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    assert cabi.filterinterp_forward_ori(ctx, flow, filt, out) == 0
lib.vfi_dev_fi_stamps(buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    assert cabi.filterinterp_forward_ori(ctx, flow, filt, out) == 0
e1.record()
torch.cuda.synchronize()
lib.vfi_dev_fi_stamps(buf)
v = list(buf)
steps = max(1, v[4])
tot = sum(v[:4])
print("%s: %.1f us per launch; per wave-step (ticks): issue %.0f  compute %.0f  vmcnt wait %.0f  barrier %.0f  = %.0f  (%d wave-steps per launch)"
      % (args.flow, e0.elapsed_time(e1) / n * 1e3, v[0] / steps, v[1] / steps, v[2] / steps, v[3] / steps, tot / steps, steps // n))
