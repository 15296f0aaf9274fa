#!/usr/bin/env python3
"""Per-op timing on the GPU (HIP events on the launch stream), for kernel development.

    python tools/bench_ops.py [--ops fi196,fi3,fi196h,fitypes,proj,dproj,corr,corr16,glue] [--flows smooth,quarter] [--iters 20]
Prints one line per (op, flow): mean ms and algorithmic GB/s (SURVEY.md 8d byte counts).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
if "--lib" in sys.argv:         # a development build of the library (make OUT=../lib_dev EXTRA=-DVFI_DEV): needed by --knobs
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from vfidkr_amd import cabi, synthetic as S  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="fi196,fi3,proj,dproj,corr")
    ap.add_argument("--flows", default="smooth,quarter")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--knobs", default="", help="comma list of flags:groups pairs for vfi_dev_filterinterp, e.g. 0:0,1:0,0:1")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    h, w = S.padded_size(args.height, args.width)
    px = h * w
    gen = S.generator()
    frame = S.frames(1, h, w, gen).to(dev)
    ctx = S.context(1, 196, h, w, gen).to(dev)
    filt = S.filters(1, h, w, gen).to(dev)
    depth = S.depth_weight(1, h, w, gen).to(dev)
    out196, out3 = torch.empty_like(ctx), torch.empty_like(frame)
    count = torch.zeros((1, 1, h, w), device=dev)
    proj = torch.zeros((1, 2, h, w), device=dev)
    ops = args.ops.split(",")
    for model in args.flows.split(","):
        flow = S.flow(1, h, w, 8.0 * w / 1984.0, gen, model).to(dev)
        if args.knobs:       # the first timed launches of a process run ~8 % slow (clocks): spend them here
            timed(lambda: cabi.filterinterp_forward_ori(ctx, flow, filt, out196), args.iters)
        for knob in [k for k in args.knobs.split(",") if k]:
            fl, gr = (int(v, 0) for v in knob.split(":"))
            cabi.lib().vfi_dev_filterinterp(fl, gr)
            ms = timed(lambda: cabi.filterinterp_forward_ori(ctx, flow, filt, out196), args.iters)
            ms3 = timed(lambda: cabi.filterinterp_forward_ori(frame, flow, filt, out3), args.iters * 5)
            print("knob flags=%#x groups=%d %-8s fi196 %8.4f ms %7.1f GB/s | fi3 %8.4f ms %7.1f GB/s"
                  % (fl, gr, model, ms, 1640.0 * px / ms / 1e6, ms3, 96.0 * px / ms3 / 1e6), flush=True)
            cabi.lib().vfi_dev_filterinterp(0, 0)
        for direct in (False, True):
            tag = "direct" if direct else "lds"
            if "fi196" in ops:
                ms = timed(lambda: cabi.filterinterp_forward_ori(ctx, flow, filt, out196, direct=direct), args.iters)
                print("fi196 %-8s %-6s %8.4f ms %8.1f GB/s" % (model, tag, ms, 1640.0 * px / ms / 1e6), flush=True)
            if "fi3" in ops:
                ms = timed(lambda: cabi.filterinterp_forward_ori(frame, flow, filt, out3, direct=direct), args.iters * 5)
                print("fi3   %-8s %-6s %8.4f ms %8.1f GB/s" % (model, tag, ms, 96.0 * px / ms / 1e6), flush=True)

        if "fi196h" in ops:
            ctx16, out16 = ctx.to(torch.float16), torch.empty_like(ctx, dtype=torch.float16)
            for direct in (False, True):
                ms = timed(lambda: cabi.filterinterp_forward_ori_f16(ctx16, flow, filt, out16, direct=direct), args.iters)
                print("fi196 fp16 storage %-8s %-6s %8.4f ms %8.1f GB/s (856 B/px)"
                      % (model, "direct" if direct else "lds", ms, 856.0 * px / ms / 1e6), flush=True)
            del ctx16, out16

        def fp():
            cabi.flowprojection_forward(flow, count, proj, 1)

        def dfp():
            cabi.depthflowprojection_forward(flow, depth, count, proj, 1)
        if "proj" in ops:
            ms = timed(fp, args.iters * 2)
            print("proj  %-8s        %8.4f ms %8.1f GB/s (all launches)" % (model, ms, 20.0 * px / ms / 1e6), flush=True)
        if "dproj" in ops:
            ms = timed(dfp, args.iters * 2)
            print("dproj %-8s        %8.4f ms %8.1f GB/s (all launches)" % (model, ms, 24.0 * px / ms / 1e6), flush=True)
    # correlation kernel-selection experiments: need a development build of the library (--lib, make EXTRA=-DVFI_DEV)
    def corr_knobs(big, flat, rows2):
        import ctypes
        fn = cabi.lib().vfi_dev_correlation
        fn.argtypes = [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int]
        fn(big, flat, rows2)

    def corr_sweep(tag):
        for a, b in S.correlation_features(1, h, w, S.generator()):
            a, b = a.to(dev), b.to(dev)
            ms = timed(lambda: cabi.correlation_forward(a, b, 4, 1, 4, 1, 1), args.iters * 2)
            print("corr %-28s C=%-3d %4dx%-4d %8.4f ms" % (tag, a.shape[1], a.shape[2], a.shape[3], ms), flush=True)
    for fthr in ([0, 64, 256, 1024] if "corrflat" in ops else []):
        corr_knobs(256, fthr, 1)
        corr_sweep("flat<%d" % fthr)
    for r2, thr in ([(1, 256), (1, 1 << 40), (0, 256), (0, 1 << 40)] if "corrrows2" in ops else []):
        corr_knobs(thr, 64, r2)
        corr_sweep("rows2=%d big>=%d" % (r2, thr))
    if "corrflat" in ops or "corrrows2" in ops:
        corr_knobs(256, 64, 1)
    if "glue" in ops:
        # the steps either side of the ops: fused launch vs the torch ops the reference uses
        import torch.nn.functional as F
        from vfidkr_amd import fused
        flow_q = (torch.randn((1, 2, h // 4, w // 4), generator=gen) * 0.4).to(dev)
        up = torch.empty((1, 2, h, w), device=dev)
        timed(lambda: cabi.filterinterp_forward_ori(ctx, S.flow(1, h, w, 8.0, gen, "smooth").to(dev), filt, out196), 20)   # clocks up
        ms = timed(lambda: cabi.flow_upsample4(flow_q, up, 20.0, 0.5), args.iters * 5)
        ms_t = timed(lambda: F.interpolate(20.0 * flow_q * 0.5, scale_factor=4, mode="bilinear"), args.iters * 5)
        print("glue  upsample4 x20 x t        %8.4f ms %8.1f GB/s | torch mul, mul, interpolate %8.4f ms"
              % (ms, 8.5 * px / ms / 1e6, ms_t), flush=True)
        ms_u = timed(lambda: (cabi.flow_upsample4(flow_q, up, 20.0, 0.5),
                              cabi.depthflowprojection_forward(up, depth, count, proj, 1)), args.iters * 2)
        ms_f = timed(lambda: cabi.depthflowprojection_forward_up4(flow_q, depth, count, proj, 20.0, 0.5, 1), args.iters * 2)
        print("glue  upsample + dproj         %8.4f ms unfused | %8.4f ms fused (%5.1f GB/s of 16.5 B/px)"
              % (ms_u, ms_f, 16.5 * px / ms_f / 1e6), flush=True)
        flow = S.flow(1, h, w, 8.0 * w / 1984.0, gen, "smooth").to(dev)
        flow2 = (-flow).contiguous()
        frame2 = S.frames(1, h, w, gen).to(dev)
        b_, o0, o2 = torch.empty_like(frame), torch.empty_like(frame), torch.empty_like(frame)

        def unfused():
            cabi.filterinterp_forward_ori(frame, flow, filt, o0)
            cabi.filterinterp_forward_ori(frame2, flow2, filt, o2)
            return o0 * 0.75 + o2 * 0.25
        ms_u = timed(unfused, args.iters * 5)
        ms_f = timed(lambda: cabi.filterinterp_blend_forward(frame, frame2, flow, flow2, filt, filt, b_, o0, o2, 0.75, 0.25),
                     args.iters * 5)
        print("glue  2 x FI(C=3) + blend      %8.4f ms unfused | %8.4f ms fused (%5.1f GB/s of 204 B/px)"
              % (ms_u, ms_f, 204.0 * px / ms_f / 1e6), flush=True)
        tot_f = tot_t = 0.0
        for a, b2 in S.correlation_features(1, h, w, gen):
            a, b2 = a.to(dev), b2.to(dev)
            fl = (torch.randn((1, 2, a.shape[2], a.shape[3]), generator=gen) * 2).to(dev)
            wo = torch.empty_like(b2)
            tot_f += timed(lambda: cabi.pwc_warp_forward(b2, fl, wo, True), args.iters * 5)

            def torch_warp():
                B, C, H, W = b2.shape
                xx = torch.arange(0, W, device=dev).view(1, 1, 1, W).expand(B, 1, H, W)
                yy = torch.arange(0, H, device=dev).view(1, 1, H, 1).expand(B, 1, H, W)
                vg = torch.cat((xx, yy), 1).float() + fl
                vg = torch.stack((2.0 * vg[:, 0] / max(W - 1, 1) - 1.0, 2.0 * vg[:, 1] / max(H - 1, 1) - 1.0), 3)
                o = F.grid_sample(b2, vg, align_corners=True)
                m = F.grid_sample(torch.ones_like(b2), vg, align_corners=True)
                m[m < 0.9999] = 0
                m[m > 0] = 1
                return o * m
            tot_t += timed(torch_warp, args.iters * 5)
        print("glue  PWC warp, 5 levels       %8.4f ms | torch grid_sample x2 + mask ops %8.4f ms" % (tot_f, tot_t), flush=True)
        # one PWC direction: coarsest level plain correlation, four levels warp -> correlation; two launches each vs one
        t2 = t1 = 0.0
        feats = [(a.to(dev), b2.to(dev)) for a, b2 in S.correlation_features(1, h, w, gen)]
        for li, (a, b2) in enumerate(feats):
            if li == 0:
                ms0 = timed(lambda: cabi.correlation_forward(a, b2, 4, 1, 4, 1, 1), args.iters * 5)
                t2 += ms0
                t1 += ms0
                continue
            fl = (torch.randn((1, 2, a.shape[2], a.shape[3]), generator=gen) * 1.5).to(dev)
            wo = torch.empty_like(b2)

            def two():
                cabi.pwc_warp_forward(b2, fl, wo, True)
                return cabi.correlation_forward(a, wo, 4, 1, 4, 1, 1)
            m2 = timed(two, args.iters * 5)
            m1 = timed(lambda: cabi.pwc_warp_correlation_forward(a, b2, fl, True), args.iters * 5)
            print("glue  level C=%-3d %4dx%-4d warp + correlation %8.4f ms | fused %8.4f ms" % (a.shape[1], a.shape[2], a.shape[3], m2, m1), flush=True)
            t2 += m2
            t1 += m1
        print("glue  one PWC direction (5 correlations, 4 warps): %8.4f ms in 9 launches | %8.4f ms in 5" % (t2, t1), flush=True)
        u8 = torch.randint(0, 256, (1, args.height, args.width, 3), dtype=torch.uint8, generator=gen).to(dev)
        pad = fused.padding_for(args.height, args.width)
        x = torch.empty((1, 3, h, w), device=dev)
        ms1 = timed(lambda: cabi.frame_u8_to_planar(u8, x, *pad), args.iters * 5)
        back = torch.empty_like(u8)
        ms2 = timed(lambda: cabi.planar_to_frame_u8(x, back, pad[2], pad[0]), args.iters * 5)
        sums = torch.zeros(2, dtype=torch.int64, device=dev)
        ms3 = timed(lambda: cabi.frame_error_sums(u8, back, sums), args.iters * 5)
        print("glue  frame in / out / error   %8.4f / %8.4f / %8.4f ms" % (ms1, ms2, ms3), flush=True)
    if "corr" in ops:
        tot_ms, tot_b = 0.0, 0.0
        for a, b in S.correlation_features(1, h, w, gen):
            a, b = a.to(dev), b.to(dev)
            ms = timed(lambda: cabi.correlation_forward(a, b, 4, 1, 4, 1, 1), args.iters * 2)
            nb = (2 * a.shape[1] + 81) * 4.0 * a.shape[2] * a.shape[3]
            tot_ms += ms
            tot_b += nb
            print("corr  C=%-3d %4dx%-4d   %8.4f ms %8.1f GB/s" % (a.shape[1], a.shape[2], a.shape[3], ms, nb / ms / 1e6), flush=True)
        print("corr  5 levels         %8.4f ms %8.1f GB/s" % (tot_ms, tot_b / tot_ms / 1e6))
    if "fitypes" in ops:
        # the C=196 launch against what its windows have to fetch: no windows at all (every pixel invalid: copy-through),
        # the smallest window (zero flow), constant flows with different alignments, smooth fields of growing magnitude
        def const(fx, fy):
            f = torch.empty((1, 2, h, w), device=dev)
            f[:, 0], f[:, 1] = fx, fy
            return f
        cases = [("invalid (copy-through)", const(1.0e6, 1.0e6)), ("zero", const(0.0, 0.0)), ("const +5.5,+3.25", const(5.5, 3.25)),
                 ("const +32.5,+0.25", const(32.5, 0.25)), ("const +0.5,+16.25", const(0.5, 16.25))]
        for sg in (2.0, 8.0):
            cases.append(("smooth sigma %g" % sg, S.flow(1, h, w, sg * w / 1984.0, gen, "smooth").to(dev)))
        # separate "bigger window" from "lanes read different windows": one outlier pixel per tile widens the bounding box
        # while every other lane still reads consecutive addresses; 0/1-pixel jitter keeps the box small but scatters lanes
        f = const(0.0, 0.0)
        f[:, 0, 0::16, 0::64], f[:, 1, 0::16, 0::64] = 9.0, 5.0
        cases.append(("zero + outlier(+9,+5)/tile", f))
        f = const(0.0, 0.0)
        f[:, 0, 0::16, 0::64], f[:, 1, 0::16, 0::64] = 18.0, 10.0
        cases.append(("zero + outlier(+18,+10)/tile", f))
        g = torch.Generator(device="cpu").manual_seed(5)
        jx = torch.randint(0, 2, (1, 1, h, w), generator=g).float().to(dev)
        jy = torch.randint(0, 2, (1, 1, h, w), generator=g).float().to(dev)
        cases.append(("jitter x in {0,1}", torch.cat([jx, jx * 0], 1).contiguous()))
        cases.append(("jitter y in {0,1}", torch.cat([jy * 0, jy], 1).contiguous()))
        cases.append(("jitter x,y in {0,1}", torch.cat([jx, jy], 1).contiguous()))
        j4 = torch.randint(0, 8, (1, 2, h, w), generator=g).float().to(dev)
        cases.append(("jitter x,y in {0..7}", j4))
        for name, f in cases:
            ms = timed(lambda: cabi.filterinterp_forward_ori(ctx, f, filt, out196), args.iters)
            print("fitypes %-24s %8.4f ms %8.1f GB/s" % (name, ms, 1640.0 * px / ms / 1e6), flush=True)
    if "corr16" in ops:
        # half storage: the tiled kernel against the one-thread-per-output kernel (taken for maps that are not 8-byte aligned)
        tot, tot_plain = 0.0, 0.0
        for a, b in S.correlation_features(1, h, w, gen):
            a, b = a.to(dev).half(), b.to(dev).half()
            ms = timed(lambda: cabi.correlation_forward(a, b, 4, 1, 4, 1, 1), args.iters * 2)
            n = a.numel()
            u1 = torch.empty(n + 1, dtype=torch.float16, device=dev)[1:].view(a.shape)
            u2 = torch.empty(n + 1, dtype=torch.float16, device=dev)[1:].view(a.shape)
            u1.copy_(a), u2.copy_(b)
            ms_plain = timed(lambda: cabi.correlation_forward(u1, u2, 4, 1, 4, 1, 1), args.iters * 2)
            nb = (2 * a.shape[1] + 81) * 2.0 * a.shape[2] * a.shape[3]
            tot += ms
            tot_plain += ms_plain
            print("corr16 C=%-3d %4dx%-4d   %8.4f ms %8.1f GB/s | one thread per output %8.4f ms"
                  % (a.shape[1], a.shape[2], a.shape[3], ms, nb / ms / 1e6, ms_plain), flush=True)
        print("corr16 5 levels         %8.4f ms | one thread per output %8.4f ms" % (tot, tot_plain))


if __name__ == "__main__":
    main()
