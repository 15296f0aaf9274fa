#!/usr/bin/env python3
"""Round-4 kernel timing at 1080p (HIP events on the launch stream):

    python tools/bench_r4.py [--what projbatch,multi,single] [--flows smooth,quarter] [--iters 100]
projbatch  FlowProjection / DepthFlowProjection: n single calls against ONE batched call of n items (n = 1, 2, 6), hot
           (same buffers every call) and cold (rotation through > 512 MB of sets)
multi      fi_forward_ori_multi<3> and <2> on the 196-channel context tensor (three / two time offsets of a direction)
single     fi_forward_ori_lds C=196 and C=3 for reference
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
if "--lib" in sys.argv:
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from vfidkr_amd import cabi, synthetic as S  # noqa: E402

TIMES = (0.25, 0.5, 0.75)


def timed(fn, iters, nsets=1):
    for i in range(max(3, nsets)):
        fn(i % nsets)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fn(i % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="projbatch,multi,single")
    ap.add_argument("--flows", default="smooth,quarter")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--kpair", type=int, default=None, help="development build: largest K staged as pairs by the shared-window kernel (0 = never)")
    ap.add_argument("--group", type=int, default=3, help="development build: flows per shared-window launch (2 or 3)")
    args = ap.parse_args()
    if args.kpair is not None:
        cabi.lib().vfi_dev_multi(args.kpair, args.group)
    dev = torch.device("cuda:0")
    h, w = S.padded_size(args.height, args.width)
    px = h * w
    gen = S.generator()
    what = args.what.split(",")
    e = lambda *s: torch.empty(s, device=dev)          # noqa: E731
    for model in args.flows.split(","):
        base = [S.flow(1, h, w, 8.0 * w / 1984.0, gen, model) for _ in range(2)]
        flows = [(base[d] * (2.0 * t)).contiguous().to(dev) for d in range(2) for t in TIMES]      # direction-major
        depth = [S.depth_weight(1, h, w, gen).to(dev) for _ in range(2)]
        dlist = [depth[0]] * 3 + [depth[1]] * 3
        if "projbatch" in what:
            nsets = int(2 * 256e6 / (24.0 * px * 6)) + 2
            sets = [([f.clone() for f in flows], [e(1, 1, h, w) for _ in flows], [e(1, 2, h, w) for _ in flows]) for _ in range(nsets)]
            for dep in (False, True):
                name = "dproj" if dep else "proj "
                bpp = 24.0 if dep else 20.0
                for n, pick in ((1, [1]), (2, [1, 4]), (6, list(range(6)))):
                    def singles(i, pick=pick, dep=dep):
                        fl, cn, out = sets[i]
                        for k in pick:
                            if dep:
                                cabi.depthflowprojection_forward(fl[k], dlist[k], cn[k], out[k], 1)
                            else:
                                cabi.flowprojection_forward(fl[k], cn[k], out[k], 1)

                    def batched(i, pick=pick, dep=dep):
                        fl, cn, out = sets[i]
                        err = cabi.flowprojection_forward_batch([fl[k] for k in pick], [cn[k] for k in pick], [out[k] for k in pick], 1,
                                                                [dlist[k] for k in pick] if dep else None)
                        assert err == 0
                    for label, fn in (("single calls", singles), ("one batch   ", batched)):
                        hot, cold = timed(fn, args.iters, 1), timed(fn, args.iters, nsets)
                        print("%s %-8s n=%d %s  hot %7.2f us (%6.2f per item, %6.1f GB/s) | cold %7.2f us (%6.2f per item, %6.1f GB/s)"
                              % (name, model, n, label, hot, hot / n, bpp * px * n / hot / 1e3, cold, cold / n, bpp * px * n / cold / 1e3),
                              flush=True)
            del sets
        if "corr" in what:
            levels = [[(a.to(dev), b.to(dev)) for a, b in S.correlation_features(1, h, w, gen)] for _ in range(2)]

            def singles(i):
                for d in range(2):
                    for a, b in levels[d]:
                        cabi.correlation_forward(a, b, 4, 1, 4, 1, 1)

            def pairs(i):
                for (a0, b0), (a1, b1) in zip(levels[0], levels[1]):
                    cabi.correlation_forward_pair(a0, b0, a1, b1, 4, 1, 4, 1, 1)
            print("corr     10 single calls %7.1f us | 5 pair calls %7.1f us" % (timed(singles, args.iters), timed(pairs, args.iters)), flush=True)
            for li, ((a0, b0), (a1, b1)) in enumerate(zip(levels[0], levels[1])):
                one = timed(lambda i: cabi.correlation_forward(a0, b0, 4, 1, 4, 1, 1), args.iters)
                two = timed(lambda i: cabi.correlation_forward_pair(a0, b0, a1, b1, 4, 1, 4, 1, 1), args.iters)
                print("corr     level %d %s: one call %6.1f us, pair %6.1f us" % (li, tuple(a0.shape), one, two), flush=True)
        if "multi" in what or "single" in what or "frames" in what or "sched" in what:
            ctx = S.context(1, 196, h, w, gen).to(dev)
            filt = S.filters(1, h, w, gen).to(dev)
            frame = S.frames(1, h, w, gen).to(dev)
            projs = []
            for k in range(3):
                c, o = e(1, 1, h, w), e(1, 2, h, w)
                assert cabi.depthflowprojection_forward(flows[k], depth[0], c, o, 1) == 0
                projs.append(o)
            outs = [torch.empty_like(ctx) for _ in range(3)]
            if "multi" in what:
                for nt in (3, 2):
                    us = timed(lambda i: cabi.filterinterp_forward_ori_multi(ctx, projs[:nt], filt, outs[:nt]), max(5, args.iters // 5))
                    nbytes = (2 * nt + 16 + 196 + nt * 196) * 4.0 * px
                    print("multi<%d> %-8s %8.1f us = %7.1f us per output, %6.1f GB/s algorithmic (%.3f of 8 TB/s)"
                          % (nt, model, us, us / nt, nbytes / us / 1e3, nbytes / us / 1e3 / 8000.0), flush=True)
            if "single" in what:
                us = timed(lambda i: cabi.filterinterp_forward_ori(ctx, projs[1], filt, outs[0]), max(5, args.iters // 5))
                print("fi196    %-8s %8.1f us, %6.1f GB/s algorithmic (%.3f of 8 TB/s)" % (model, us, 1640.0 * px / us / 1e3, 1640.0 * px / us / 8e6), flush=True)
                o3 = torch.empty_like(frame)
                us = timed(lambda i: cabi.filterinterp_forward_ori(frame, projs[1], filt, o3), args.iters)
                print("fi3      %-8s %8.1f us, %6.1f GB/s algorithmic" % (model, us, 96.0 * px / us / 1e3), flush=True)
            if "frames" in what:
                # the three frame warps of a direction (C = 3): three single-flow launches against the shared-window entry point
                o3 = [torch.empty_like(frame) for _ in range(3)]
                us1 = timed(lambda i: [cabi.filterinterp_forward_ori(frame, projs[k], filt, o3[k]) for k in range(3)], args.iters)
                us2 = timed(lambda i: cabi.filterinterp_forward_ori_multi(frame, projs, filt, o3), args.iters)
                print("frames   %-8s three C=3 launches %7.1f us | one shared-window call (2 + 1) %7.1f us" % (model, us1, us2), flush=True)
            if "sched" in what:
                # both directions' context warps (three time offsets each) on two streams, as the best schedule runs them
                from vfidkr_amd import fused
                lanes = fused.DirectionStreams(dev)
                ctx2 = S.context(1, 196, h, w, gen).to(dev)
                outs2 = [torch.empty_like(ctx2) for _ in range(3)]

                def both(i):
                    lanes.fork()
                    with lanes.direction(0):
                        assert cabi.filterinterp_forward_ori_multi(ctx, projs, filt, outs) == 0
                    with lanes.direction(1):
                        assert cabi.filterinterp_forward_ori_multi(ctx2, projs, filt, outs2) == 0
                    lanes.join()
                us = timed(both, max(5, args.iters // 5))
                print("sched    %-8s both directions' context warps (2 x 3 outputs) on two streams: %8.1f us" % (model, us), flush=True)
                del ctx2, outs2
            del ctx, outs


if __name__ == "__main__":
    main()
