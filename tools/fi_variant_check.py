#!/usr/bin/env python3
"""Development: a kernel variant of the C=196 FilterInterpolation launch (vfi_dev_filterinterp flags, -DVFI_DEV build)
against the default kernel -- same bits on every flow model and on ragged frames, then time.
    python tools/fi_variant_check.py --lib <pkg>/lib_dev/libvfi_hip.so --flags 72[,...]
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
    ap.add_argument("--lib", required=True)
    ap.add_argument("--flags", default="72")
    ap.add_argument("--base", type=lambda v: int(v, 0), default=8)
    ap.add_argument("--channels", type=int, default=196)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--time-only", action="store_true", help="ablation flags give wrong results on purpose")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    knob = cabi.lib().vfi_dev_filterinterp
    variants = [int(v, 0) for v in args.flags.split(",")]
    gen = S.generator()
    ok = True
    for (hh, ww, C) in (() if args.time_only else ((1080, 1920, args.channels), (270, 333, 5), (64, 70, 3))):
        h, w = (S.padded_size(hh, ww) if hh == 1080 else (hh, ww))
        ctx = S.context(1, C, h, w, gen).to(dev)
        filt = S.filters(1, h, w, gen).to(dev)
        for model in ("smooth", "quarter", "uniform1", "wild"):
            flow = S.flow(1, h, w, 8.0 * w / 1984.0, gen, model).to(dev)
            knob(args.base, 0)
            ref = torch.empty_like(ctx)
            assert cabi.filterinterp_forward_ori(ctx, flow, filt, ref) == 0
            for fl in variants:
                knob(fl, 0)
                out = torch.full_like(ctx, float("nan"))
                assert cabi.filterinterp_forward_ori(ctx, flow, filt, out) == 0
                same = torch.equal(out.view(torch.int32), ref.view(torch.int32))
                ok &= same
                print("parity %4dx%-4d C=%-3d %-8s flags=%#x %s" % (h, w, C, model, fl, "same bits" if same else
                      "DIFFERENT (max abs %g)" % (out - ref).abs().nan_to_num(1e30).max().item()), flush=True)
    h, w = S.padded_size(1080, 1920)
    ctx = S.context(1, args.channels, h, w, gen).to(dev)
    filt = S.filters(1, h, w, gen).to(dev)
    out = torch.empty_like(ctx)
    for model in ("smooth", "quarter"):
        flow = S.flow(1, h, w, 8.0 * w / 1984.0, gen, model).to(dev)
        knob(args.base, 0)
        timed(lambda: cabi.filterinterp_forward_ori(ctx, flow, filt, out), args.iters)      # clocks
        for rep in range(2):
            for fl in [args.base] + variants:
                knob(fl, 0)
                ms = timed(lambda: cabi.filterinterp_forward_ori(ctx, flow, filt, out), args.iters)
                print("time %-8s flags=%#-6x %8.4f ms  frac %.3f" % (model, fl, ms, 1640.0 * h * w / ms / 1e6 / 8000.0), flush=True)
    knob(args.base, 0)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
