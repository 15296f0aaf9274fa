#!/usr/bin/env python3
"""Soak of the other hot-path entry points against the CPU oracle on random shapes (larger and more varied than the
hypothesis-drawn unit tests): correlation fp32 and half (every k = 1 kernel: flat, tiled, 16-byte rows), PWC-Net's warp
alone and feeding the correlation in one launch, the x4-upsample fused into the projection, MinDepthFlowProjection,
Interpolation, and the backward passes of FilterInterpolation, the projections, the correlation and Interpolation
(per-pixel gradients bit-exact, scattered image gradients within GRAD_TOL).
    python tests/soak_ops.py [cases] [seed]        (not collected by pytest; the oracle is the checker)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi  # noqa: E402
from oracle import cpu_oracle as oracle  # noqa: E402  (test infrastructure: the checker)
oracle.build()

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
f32 = np.float32
dev = torch.device("cuda:0")
gpu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
cpu = lambda t: t.detach().cpu().numpy()                               # noqa: E731
bad = 0


def check(name, ok, info=""):
    global bad
    if not ok:
        bad += 1
        print("MISMATCH %s %s" % (name, info), flush=True)


for it in range(cases):
    B = int(rng.choice([1, 1, 2]))
    C = int(rng.choice([1, 3, 8, 19, 32, 67]))
    H = int(rng.choice([1, 4, 9, 33, 64, 72, 130]))
    W = int(rng.choice([1, 5, 16, 31, 64, 124, 248]))
    info = "case %d B=%d C=%d H=%d W=%d" % (it, B, C, H, W)
    f1 = rng.standard_normal((B, C, H, W)).astype(f32)
    f2 = rng.standard_normal((B, C, H, W)).astype(f32)
    g1, g2 = gpu(f1), gpu(f2)
    # ---- correlation, PWC-Net's configuration
    want = oracle.correlation_fwd(f1, f2, 4, 1, 4, 1, 1, order=1, fmad=1)
    check("correlation fp32", np.array_equal(cpu(cabi.correlation_forward(g1, g2, 4, 1, 4, 1, 1)), want), info)
    if C <= 32:
        h1, h2 = f1.astype(np.float16), f2.astype(np.float16)
        want16 = oracle.correlation_fwd_f16(h1, h2, 4, 1, 4, 1, 1)
        got16 = cpu(cabi.correlation_forward(gpu(h1), gpu(h2), 4, 1, 4, 1, 1))
        check("correlation half", np.array_equal(got16.view(np.uint16), want16.view(np.uint16)), info)
    # ---- PWC warp, alone and fused into the correlation
    flo = (rng.standard_normal((B, 2, H, W)) * float(rng.choice([0.3, 2.0, 9.0]))).astype(f32)
    gfl = gpu(flo)
    warped = torch.empty_like(g2)
    assert cabi.pwc_warp_forward(g2, gfl, warped, True) == 0
    w_ref = oracle.pwc_warp(f2, flo, True, fmad=1)
    check("pwc_warp", np.array_equal(cpu(warped), w_ref), info)
    fused = cabi.pwc_warp_correlation_forward(g1, g2, gfl, True)
    check("warp + correlation, one launch", np.array_equal(cpu(fused), oracle.correlation_fwd(f1, w_ref, 4, 1, 4, 1, 1, order=1, fmad=1)), info)
    # ---- projection from the quarter-resolution flow, MinDepth, Interpolation
    hq, wq = max(1, H // 4), max(1, W // 4)
    fq = (np.round(rng.standard_normal((B, 2, hq, wq)) * 4) / 4).astype(f32)          # dyadic: x 20 x 0.25 / 0.5 stay exact
    for fh in (0, 1):
        count = torch.full((B, 1, 4 * hq, 4 * wq), float("nan"), device=dev)
        out = torch.full((B, 2, 4 * hq, 4 * wq), float("nan"), device=dev)
        assert cabi.flowprojection_forward_up4(gpu(fq), count, out, 16.0, 0.5, fh) == 0
        r, rc = oracle.flowproj_up4_fwd(fq, 16.0, 0.5, fh)
        check("projection from quarter flow", np.array_equal(cpu(count), rc) and np.abs(cpu(out) - r).max() <= 1e-4, info + " fillhole=%d" % fh)
    fd = (np.round(flo * 8) / 8).astype(f32)
    wgt = (np.round(rng.uniform(0.1, 1.0, (B, 1, H, W)) * 16) / 16 + 1 / 16).astype(f32)
    for fh in (0, 1):
        count, out = torch.zeros((B, 1, H, W), device=dev), torch.zeros((B, 2, H, W), device=dev)
        assert cabi.mindepthflowprojection_forward(gpu(fd), gpu(wgt), count, out, fh) == 0
        r, rc = oracle.mindepthflowproj_fwd(fd, wgt, fh)
        check("MinDepthFlowProjection", np.array_equal(cpu(count), rc) and np.array_equal(cpu(out), r), info + " fillhole=%d" % fh)
    if C <= 8:
        out = torch.full((B, C, H, W), float("nan"), device=dev)
        assert cabi.interpolation_forward(g1, gfl, out) == 0
        check("Interpolation", np.array_equal(cpu(out), oracle.interp_fwd(f1, flo, fmad=1)), info)
        # ---- FilterInterpolation backward
        filt = rng.random((B, 16, H, W), dtype=f32)
        gout = rng.standard_normal((B, C, H, W)).astype(f32)
        gi1, gi2, gi3 = (torch.zeros(s, device=dev) for s in ((B, C, H, W), (B, 2, H, W), (B, 16, H, W)))
        assert cabi.filterinterp_backward_ori(g1, gfl, gpu(filt), gpu(gout), gi1, gi2, gi3) == 0
        r1, r2, r3 = oracle.filterinterp_ori_bwd(f1, flo, filt, gout, fmad=1)
        check("FilterInterpolation backward", np.array_equal(cpu(gi2), r2) and np.array_equal(cpu(gi3), r3)
              and np.abs(cpu(gi1) - r1).max() <= 2e-6 * max(1.0, np.abs(r1).max()), info)
    # ---- backward passes: projection (bit-exact), correlation (bit-exact), Interpolation (image gradient within GRAD_TOL)
    dep = rng.uniform(0.1, 1.0, (B, 1, H, W)).astype(f32)
    po, pc = oracle.depthflowproj_fwd(flo, dep, 0)
    cnt = np.where(pc > 0, pc, 1).astype(f32)
    g2o = rng.standard_normal((B, 2, H, W)).astype(f32)
    gf = torch.zeros((B, 2, H, W), device=dev)
    assert cabi.flowprojection_backward(gfl, gpu(cnt), gpu(g2o), gf) == 0
    check("FlowProjection backward", np.array_equal(cpu(gf), oracle.flowproj_bwd(flo, cnt, g2o)), info)
    gf.zero_()
    gd = torch.zeros((B, 1, H, W), device=dev)
    assert cabi.depthflowprojection_backward(gfl, gpu(dep), gpu(cnt), gpu(po), gpu(g2o), gf, gd) == 0
    rf, rd = oracle.depthflowproj_bwd(flo, dep, cnt, po, g2o)
    check("DepthFlowProjection backward", np.array_equal(cpu(gf), rf) and np.array_equal(cpu(gd), rd), info)
    if C <= 19 and H * W <= 10000:
        gco = rng.standard_normal((B, 81, H, W)).astype(f32)
        c1, c2 = cabi.correlation_backward(g1, g2, gpu(gco), 4, 1, 4, 1, 1)
        r1, r2 = oracle.correlation_bwd(f1, f2, gco, 4, 1, 4, 1, 1)
        check("correlation backward", np.array_equal(cpu(c1), r1) and np.array_equal(cpu(c2), r2), info)
    if C <= 8:
        gio = rng.standard_normal((B, C, H, W)).astype(f32)
        gi, gfl2 = torch.zeros((B, C, H, W), device=dev), torch.zeros((B, 2, H, W), device=dev)
        assert cabi.interpolation_backward(g1, gfl, gpu(gio), gi, gfl2) == 0
        ri, rfl = oracle.interp_bwd(f1, flo, gio, fmad=1)
        check("Interpolation backward", np.array_equal(cpu(gfl2), rfl) and np.abs(cpu(gi) - ri).max() <= 2e-6 * max(1.0, np.abs(ri).max()), info)
    # ---- SeparableConv / SeparableConvFlow forward + backward (filter sizes 1 .. 13), deformable backward
    fs = int(rng.choice([1, 2, 5, 9, 13]))
    if C == 3 and H >= fs and W >= fs and H * W <= 20000:        # (three channels: separableconv_cuda.cc refuses anything else)
        oh, ow = H - fs + 1, W - fs + 1
        v = rng.random((B, fs, oh, ow), dtype=f32)
        hh = rng.random((B, fs, oh, ow), dtype=f32)
        gv, gh = gpu(v), gpu(hh)
        so = torch.full((B, C, oh, ow), float("nan"), device=dev)
        assert cabi.separableconv_forward(g1, gv, gh, so) == 0
        check("SeparableConv", np.array_equal(cpu(so), oracle.sepconv_fwd(f1, v, hh, fmad=1)), info + " fs=%d" % fs)
        fo = torch.full((B, 2, oh, ow), float("nan"), device=dev)
        assert cabi.separableconvflow_forward(g1, gv, gh, fo) == 0
        check("SeparableConvFlow", np.array_equal(cpu(fo), oracle.sepconvflow_fwd(v, hh, H, W, fmad=1)), info + " fs=%d" % fs)
        sgo = rng.standard_normal((B, C, oh, ow)).astype(f32)
        s1_, s2_, s3_ = torch.zeros_like(g1), torch.zeros_like(gv), torch.zeros_like(gh)
        assert cabi.separableconv_backward(g1, gv, gh, gpu(sgo), s1_, s2_, s3_) == 0
        q1, q2, q3 = oracle.sepconv_bwd(f1, v, hh, sgo)
        check("SeparableConv backward", np.array_equal(cpu(s1_), q1) and np.array_equal(cpu(s2_), q2) and np.array_equal(cpu(s3_), q3), info + " fs=%d" % fs)
    if C <= 3 and H * W <= 6000:
        filt = rng.random((B, 16, H, W), dtype=f32)
        off = (rng.standard_normal((B, 32, H, W)) * 0.7).astype(f32)
        dgo = rng.standard_normal((B, C, H, W)).astype(f32)
        for variant in (0, 1, 2):
            d1, d2 = torch.zeros((B, C, H, W), device=dev), torch.zeros((B, 2, H, W), device=dev)
            do = torch.zeros((B, 32, H, W), device=dev)
            if variant == 2:
                err = cabi.filterinterp_backward_defor(variant, g1, gfl, gpu(off), None, gpu(dgo), d1, d2, do, None)
                dfl = None
            else:
                dfl = torch.zeros((B, 16, H, W), device=dev)
                err = cabi.filterinterp_backward_defor(variant, g1, gfl, gpu(filt), gpu(off), gpu(dgo), d1, d2, dfl, do)
            assert err == 0
            e1, e2, e3, e4 = oracle.filterinterp_defor_bwd(variant, f1, flo, filt, off, dgo, fmad=1)
            okb = (np.abs(cpu(d1) - e1).max() <= 2e-6 * max(1.0, np.abs(e1).max()) and np.array_equal(cpu(d2), e2) and np.array_equal(cpu(do), e4)
                   and (variant == 2 or np.array_equal(cpu(dfl), e3)))
            check("deformable backward %d" % variant, okb, info)
    if it % 20 == 19:
        print("%d cases, %d mismatches" % (it + 1, bad), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
