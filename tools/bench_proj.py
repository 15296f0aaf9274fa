#!/usr/bin/env python3
"""FlowProjection / DepthFlowProjection timing, cache-hot and cache-cold (HIP events on the launch stream).

    python tools/bench_proj.py [--flows smooth,quarter,uniform1] [--iters 200] [--height 1080 --width 1920]
hot  = the same buffers every call (45.7 MB: inside the 256 MB Infinity Cache);
cold = calls rotate through enough distinct (flow, depth, count, out) sets to exceed 2 x 256 MB, so no
       line of a call's working set is on the die when it starts.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
if "--lib" in sys.argv:         # a development build of the library (make OUT=... EXTRA=...)
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from vfidkr_amd import cabi, synthetic as S  # noqa: E402


def timed(fn, iters, nsets):
    for i in range(max(3, nsets)):
        fn(i % nsets)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        fn(i % nsets)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flows", default="smooth,quarter,uniform1")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--fillhole", type=int, default=1)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    h, w = S.padded_size(args.height, args.width)
    px = h * w
    gen = S.generator()
    nsets = max(2, int(2 * 256e6 / (24.0 * px)) + 1)
    for model in args.flows.split(","):
        base = S.flow(1, h, w, 8.0 * w / 1984.0, gen, model)
        flows = [base.clone().to(dev) for _ in range(nsets)]
        depths = [S.depth_weight(1, h, w, gen).to(dev) for _ in range(nsets)]
        counts = [torch.empty((1, 1, h, w), device=dev) for _ in range(nsets)]
        outs = [torch.empty((1, 2, h, w), device=dev) for _ in range(nsets)]
        for name, nbytes, fn in (
            ("proj ", 20.0, lambda i: cabi.flowprojection_forward(flows[i], counts[i], outs[i], args.fillhole)),
            ("dproj", 24.0, lambda i: cabi.depthflowprojection_forward(flows[i], depths[i], counts[i], outs[i], args.fillhole)),
        ):
            hot = timed(fn, args.iters, 1)
            cold = timed(fn, args.iters, nsets)
            print("%s %-8s %dx%d  hot %7.2f us %7.1f GB/s | cold %7.2f us %7.1f GB/s (%d sets)"
                  % (name, model, h, w, hot, nbytes * px / hot / 1e3, cold, nbytes * px / cold / 1e3, nsets), flush=True)
        del flows, depths, counts, outs


if __name__ == "__main__":
    main()
