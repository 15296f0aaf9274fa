#!/usr/bin/env python3
"""Times the ops outside the DAIN hot path (SURVEY A5-A7 and the training kernels) at a padded 1080p frame.
python tools/bench_rest.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

H, W = S.padded_size(1080, 1920)
gen = S.generator()
img = S.frames(1, H, W, gen).cuda()
flow = S.flow(1, H, W, 8.0, gen, "smooth").cuda()
filt = S.filters(1, H, W, gen).cuda()
px = H * W


def timed(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = torch.empty_like(img)
print("interpolation fwd C=3          %8.4f ms" % timed(lambda: cabi.interpolation_forward(img, flow, out)), flush=True)
gi, gf = torch.zeros_like(img), torch.zeros_like(flow)
gout = torch.randn(img.shape, generator=gen).cuda()
print("interpolation bwd C=3          %8.4f ms" % timed(lambda: cabi.interpolation_backward(img, flow, gout, gi, gf)), flush=True)
for fs in (5, 25, 51):
    oh, ow = H - fs + 1, W - fs + 1
    v = torch.rand((1, fs, oh, ow), generator=gen).cuda()
    hh = torch.rand((1, fs, oh, ow), generator=gen).cuda()
    o = torch.empty((1, 3, oh, ow), device="cuda")
    ms = timed(lambda: cabi.separableconv_forward(img, v, hh, o), 3)
    print("separableconv fwd fs=%-2d        %8.4f ms  (%.1f GMAC/s)" % (fs, ms, 3.0 * fs * fs * oh * ow / ms / 1e6), flush=True)
    fo = torch.empty((1, 2, oh, ow), device="cuda")
    print("separableconvflow fwd fs=%-2d    %8.4f ms" % (fs, timed(lambda: cabi.separableconvflow_forward(img, v, hh, fo), 3)), flush=True)
g1, g2, g3 = torch.zeros_like(img), torch.zeros_like(flow), torch.zeros_like(filt)
print("filterinterp bwd ori C=3       %8.4f ms" % timed(lambda: cabi.filterinterp_backward_ori(img, flow, filt, gout, g1, g2, g3), 5), flush=True)
count, proj = torch.empty((1, 1, H, W), device="cuda"), torch.empty_like(flow)
cabi.flowprojection_forward(flow, count, proj, 0)
gp = torch.zeros_like(flow)
gpo = torch.randn(flow.shape, generator=gen).cuda()
print("flowprojection bwd             %8.4f ms" % timed(lambda: cabi.flowprojection_backward(flow, count, gpo, gp)), flush=True)
f1, f2 = S.correlation_features(1, H, W, gen)[-1]
f1, f2 = f1.cuda(), f2.cuda()
co = cabi.correlation_forward(f1, f2, 4, 1, 4, 1, 1)
gco = torch.randn(co.shape, generator=gen).cuda()
print("correlation bwd C=%d %dx%d  %8.4f ms" % (f1.shape[1], f1.shape[2], f1.shape[3],
                                                 timed(lambda: cabi.correlation_backward(f1, f2, gco, 4, 1, 4, 1, 1), 5)), flush=True)
wd = torch.rand((1, 1, H, W), generator=gen).cuda() + 0.1
c2, o2 = torch.zeros((1, 1, H, W), device="cuda"), torch.zeros_like(flow)


def md():
    c2.zero_(), o2.zero_()
    cabi.mindepthflowprojection_forward(flow, wd, c2, o2, 1)


print("mindepthflowprojection fwd     %8.4f ms (incl. two memsets)" % timed(md), flush=True)
