#!/usr/bin/env python3
"""DepthFlowProjection at 1080p against the spread of the depth weights: a tile whose weights span more than 64x is
accumulated in more than one pass (weight classes)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, vfidkr_amd
from vfidkr_amd import cabi, synthetic as S
dev = torch.device("cuda:0")
h, w = S.padded_size(1080, 1920)
gen = S.generator()
flow = S.flow(1, h, w, 8.0, gen, "smooth").to(dev)
count = torch.empty(1, 1, h, w, device=dev); out = torch.empty(1, 2, h, w, device=dev)
def timed(fn, n=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, d in (("U(0.1,1)", S.depth_weight(1, h, w, gen)),
                ("1e-6+exp(-smooth d in 0..6)", 1e-6 + torch.exp(-(torch.nn.functional.interpolate(torch.rand(1, 1, 10, 17, generator=gen) * 6, size=(h, w), mode="bilinear")))),
                ("1e-6+exp(-U(0,6)) per pixel", 1e-6 + torch.exp(-torch.rand(1, 1, h, w, generator=gen) * 6)),
                ("1e-6+exp(-U(0,12)) per pixel", 1e-6 + torch.exp(-torch.rand(1, 1, h, w, generator=gen) * 12))):
    dd = d.contiguous().to(dev)
    print("%-34s %7.1f us" % (name, timed(lambda: cabi.depthflowprojection_forward(flow, dd, count, out, 1))), flush=True)
