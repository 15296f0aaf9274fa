#!/usr/bin/env python3
"""Which two of a direction's three time offsets should share a shared-window FilterInterpolation launch?
    python tools/fi_pairing.py
The library pairs the flows in the order given (first two together, the third alone); measured at 1152x1984, C = 196:
(0.25, 0.5) + 0.75 beats the other two pairings by 5 % (smooth field) to 11 % (quarter field): the two smallest windows share best."""
import os, sys
sys.path.insert(0, '/root/repo')
import torch
import vfidkr_amd
from vfidkr_amd import cabi, synthetic as S
sys.path.insert(0, '/root/repo/tools')
dev = torch.device('cuda:0')
h, w = S.padded_size(1080, 1920)
gen = S.generator()
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for model in ('smooth', 'quarter'):
    base = S.flow(1, h, w, 8.0 * w / 1984.0, gen, model)
    depth = S.depth_weight(1, h, w, gen).to(dev)
    ctx = S.context(1, 196, h, w, gen).to(dev)
    filt = S.filters(1, h, w, gen).to(dev)
    projs = []
    for t in (0.25, 0.5, 0.75):
        c, o = torch.empty((1, 1, h, w), device=dev), torch.empty((1, 2, h, w), device=dev)
        assert cabi.depthflowprojection_forward((base * (2.0 * t)).contiguous().to(dev), depth, c, o, 1) == 0
        projs.append(o)
    outs = [torch.empty_like(ctx) for _ in range(3)]
    for name, order in (('(0.25,0.5)+0.75', [0, 1, 2]), ('(0.5,0.75)+0.25', [1, 2, 0]), ('(0.25,0.75)+0.5', [0, 2, 1])):
        p = [projs[i] for i in order]
        us = timed(lambda: cabi.filterinterp_forward_ori_multi(ctx, p, filt, outs))
        us2 = timed(lambda: cabi.filterinterp_forward_ori_multi(ctx, p[:2], filt, outs[:2]))
        us1 = timed(lambda: cabi.filterinterp_forward_ori(ctx, p[2], filt, outs[2]))
        print('%-8s %-18s three flows %7.1f us (pair %7.1f + single %7.1f)' % (model, name, us, us2, us1), flush=True)
