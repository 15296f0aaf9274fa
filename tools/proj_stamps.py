#!/usr/bin/env python3
"""Timeline of proj_pull's workgroups from a -DPROJ_STAMPS development build of the library.

    make -C <pkg>/csrc OUT=../lib_vstamp EXTRA=-DPROJ_STAMPS
    python tools/proj_stamps.py --lib <pkg>/lib_vstamp/libvfi_hip.so
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402
vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

dev = torch.device("cuda:0")
h, w = S.padded_size(1080, 1920)
gen = S.generator()
flow = S.flow(1, h, w, 8.0, gen, "smooth").to(dev)
count = torch.empty((1, 1, h, w), device=dev)
out = torch.empty((1, 2, h, w), device=dev)
for _ in range(50):
    cabi.flowprojection_forward(flow, count, out, 1)
torch.cuda.synchronize()
nt = ((h + 15) // 16) * ((w + 63) // 64)
buf = np.zeros((nt, 8), np.uint64)
fn = cabi.lib().vfi_dev_projection_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
assert fn(1, h, w, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), buf.ctypes.data_as(ctypes.c_void_p)) == 0
t0, t1, t2, t3, r0, r1 = (buf[:, i].astype(np.float64) for i in range(6))
base = r0.min()
print("tiles", nt, " kernel span (100 MHz clock): %.2f us" % ((r1.max() - base) / 100.0))
print("shader clock / real clock: %.3f GHz" % (((t3 - t0).sum() / (r1 - r0).sum()) * 0.1))
start = (r0 - base) / 100.0
end = (r1 - base) / 100.0
print("start  min %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (start.min(), np.median(start), np.percentile(start, 90), start.max()))
print("end    min %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (end.min(), np.median(end), np.percentile(end, 90), end.max()))
life = end - start
print("life   min %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (life.min(), np.median(life), np.percentile(life, 90), life.max()))
cyc = t3 - t0
for name, a, b in (("start..accumulated", t0, t1), ("..tile stored", t1, t2), ("..end", t2, t3)):
    d = b - a
    print("%-14s median %8.0f cycles (%.0f %% of life)" % (name, np.median(d), 100.0 * np.median(d) / np.median(cyc)))
hw = buf[:, 6].astype(np.int64)
xcc = buf[:, 7].astype(np.int64) & 0xf
ta = (buf[:, 7].astype(np.int64) >> 8).astype(np.float64)
print("record arrived + LDS zeroed after: p10 %.0f p50 %.0f p90 %.0f max %.0f cycles" % (np.percentile(ta, 10), np.median(ta), np.percentile(ta, 90), ta.max()))
cu = (hw >> 8) & 0xf
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 0x1
key = xcc * 1000 + se * 100 + sh * 16 + cu
u, c = np.unique(key, return_counts=True)
print("distinct (xcc, se, sh, cu):", len(u), " workgroups per CU: min %d median %d max %d" % (c.min(), np.median(c), c.max()))
late = start > np.percentile(start, 50)
print("workgroups starting after the median start: %d, their mean start %.2f us" % (late.sum(), start[late].mean()))
order = np.argsort(start)
print("start time of workgroup #: " + "  ".join("%d: %.2f" % (i, start[order[i]]) for i in (0, 500, 1000, 1500, 2000, 2100, 2200, nt - 1)))
for name, a, b in (("start..accumulated", t0, t1), ("..tile stored", t1, t2), ("..end", t2, t3)):
    d = (b - a)
    print("%-14s cycles: p10 %7.0f p50 %7.0f p90 %7.0f max %7.0f" % (name, np.percentile(d, 10), np.median(d), np.percentile(d, 90), d.max()))
tx_n = (w + 63) // 64
slow = np.argsort(-life)[:24]
print("slowest tiles (tx,ty,life us, xcc): " + " ".join("(%d,%d,%.1f,%d)" % (i % tx_n, i // tx_n, life[i], xcc[i]) for i in slow))
for q in range(8):
    m = xcc == q
    print("xcc %d: n=%d life p50 %.2f max %.2f  end max %.2f" % (q, m.sum(), np.median(life[m]), life[m].max(), end[m].max()))
# per CU: number of WGs vs mean life
for n in np.unique(c):
    cus = u[c == n]
    m = np.isin(key, cus)
    print("CUs with %d workgroups: %d, life p50 %.2f max %.2f" % (n, len(cus), np.median(life[m]), life[m].max()))
ty = np.arange(nt) // tx_n
print("life by tile row (median): " + " ".join("%.1f" % np.median(life[ty == r]) for r in range(0, ty.max() + 1, 6)))
