#!/usr/bin/env python3
"""A/B of the correlation kernels at PWC-Net's configuration: the tiled vector kernel (corr_forward_k1_rows2) against the
matrix-core kernel (corr_forward_k1_mfma) of a -DVFI_DEV build -- same bits?, time per level, HIP events.
    tools/mkvariant.sh corrdev correlation.hip -DVFI_DEV
    python tools/corr_mfma_ab.py --lib <pkg>/lib_vcorrdev/libvfi_hip.so [--height 1080 --width 1920]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--iters", type=int, default=200)
args = ap.parse_args()
vfidkr_amd.LIB_PATH = os.path.abspath(args.lib)
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

lib = cabi.lib()
lib.vfi_dev_correlation_mfma.argtypes = [ctypes.c_int]
lib.vfi_dev_correlation_mfma.restype = None
dev = torch.device("cuda:0")
h, w = S.padded_size(args.height, args.width)
gen = S.generator()


def timed(fn):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters * 1e3


tot = [0.0, 0.0]
for f1, f2 in S.correlation_features(1, h, w, gen):
    a, b = f1.to(dev), f2.to(dev)
    res = []
    for k in (0, 1):
        lib.vfi_dev_correlation_mfma(k)
        out = cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)
        us = timed(lambda: cabi.correlation_forward(a, b, 4, 1, 4, 1, 1))
        res.append((out, us))
        tot[k] += us
    nbytes = (2 * a.shape[1] + 81) * 4.0 * a.shape[2] * a.shape[3]
    print("C=%3d %4dx%-4d  vector %7.2f us (%6.1f GB/s)  matrix cores %7.2f us (%6.1f GB/s)  same bits: %s"
          % (a.shape[1], a.shape[2], a.shape[3], res[0][1], nbytes / res[0][1] / 1e3, res[1][1], nbytes / res[1][1] / 1e3,
             bool(torch.equal(res[0][0], res[1][0]))), flush=True)
print("five levels: vector %.1f us, matrix cores %.1f us" % (tot[0], tot[1]))
