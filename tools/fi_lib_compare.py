#!/usr/bin/env python3
"""Development: the C=196 / C=3 FilterInterpolation outputs of a variant build of the library against the product build's
(two processes: a process loads one library).
    python tools/fi_lib_compare.py --save /tmp/ref.pt                    (product library)
    python tools/fi_lib_compare.py --lib <pkg>/lib_vX/libvfi_hip.so --check /tmp/ref.pt
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--save", default=None)
    ap.add_argument("--check", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = S.generator()
    outs = {}
    for (hh, ww, C) in ((1080, 1920, 24), (270, 333, 5), (64, 70, 3)):
        h, w = (S.padded_size(hh, ww) if hh == 1080 else (hh, ww))
        ctx = S.context(1, C, h, w, gen).to(dev)
        filt = S.filters(1, h, w, gen).to(dev)
        for model in ("smooth", "quarter", "uniform1", "wild"):
            flow = S.flow(1, h, w, 8.0 * w / 1984.0, gen, model).to(dev)
            out = torch.full_like(ctx, float("nan"))
            assert cabi.filterinterp_forward_ori(ctx, flow, filt, out) == 0
            outs["%dx%d_%d_%s" % (h, w, C, model)] = out.cpu()
    if args.save:
        torch.save(outs, args.save)
        print("saved", len(outs), "outputs")
    if args.check:
        ref = torch.load(args.check)
        bad = [k for k in outs if not torch.equal(outs[k].view(torch.int32), ref[k].view(torch.int32))]
        print("same bits on all %d cases" % len(outs) if not bad else "DIFFERENT: %s" % bad)
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
