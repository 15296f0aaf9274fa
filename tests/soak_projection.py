#!/usr/bin/env python3
"""Soak: FlowProjection / DepthFlowProjection (pull kernels, fallback included) against the CPU oracle on many random
frames.  Dyadic inputs (multiples of 1/8, weights multiples of 1/16): every sum is exact in any order, so count and flow
must equal the oracle's bit for bit; non-dyadic inputs: within 1e-4 and identical from run to run.
    python tests/soak_projection.py [cases] [seed]        (not collected by pytest; the oracle is the checker, so it lives under tests/)
    --lib <path> picks a development build of the library
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
if "--lib" in sys.argv:         # a development build of the library (tools/mkvariant.sh)
    _i = sys.argv.index("--lib")
    vfidkr_amd.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from vfidkr_amd import cabi  # noqa: E402
from oracle import cpu_oracle as oracle  # noqa: E402  (test infrastructure: the checker)
oracle.build()

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
f32 = np.float32
dev = torch.device("cuda:0")
bad = 0
for it in range(cases):
    B = int(rng.choice([1, 1, 2]))
    H = int(rng.choice([1, 5, 16, 17, 33, 64, 100, 211, 400]))
    W = int(rng.choice([1, 3, 63, 64, 65, 130, 256, 517, 900]))
    kind = rng.choice(["smooth", "rough", "wild", "converge", "zero"])
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    if kind == "smooth":
        a, b = rng.uniform(-0.05, 0.05, 2)
        flow = np.stack([a * xs + 3.0, b * ys - 2.0])[None].repeat(B, 0)
    elif kind == "rough":
        flow = rng.normal(0, float(rng.choice([1.0, 4.0, 12.0])), (B, 2, H, W))
    elif kind == "wild":
        flow = rng.uniform(-W / 2, W / 2, (B, 2, H, W))
    elif kind == "converge":                            # everything lands near one point: busy cells (the rescaling pass)
        flow = np.stack([(W // 2 - xs) * 0.9, (H // 2 - ys) * 0.9])[None].repeat(B, 0)
    else:
        flow = np.zeros((B, 2, H, W))
    dyadic = rng.random() < 0.7
    flow = (np.round(flow * 8) / 8 if dyadic else flow).astype(f32)
    wgt = (np.round(rng.uniform(0.1, 1.0, (B, 1, H, W)) * 16) / 16 + 1 / 16).astype(f32)
    if not dyadic and rng.random() < 0.5:
        wgt = (wgt * np.exp(rng.uniform(-12, 0, (B, 1, H, W)))).astype(f32)          # wide weight range: the class passes
    gf, gw = torch.from_numpy(flow).to(dev), torch.from_numpy(wgt).to(dev)
    ok = True
    for fh in (0, 1):
        for depth in (False, True):
            count = torch.full((B, 1, H, W), float("nan"), device=dev)
            out = torch.full((B, 2, H, W), float("nan"), device=dev)
            if depth:
                assert cabi.depthflowprojection_forward(gf, gw, count, out, fh) == 0
                r, rc = oracle.depthflowproj_fwd(flow, wgt, fh)
            else:
                assert cabi.flowprojection_forward(gf, count, out, fh) == 0
                r, rc = oracle.flowproj_fwd(flow, fh)
            c, o = count.cpu().numpy(), out.cpu().numpy()
            if dyadic:
                good = np.array_equal(c, rc) and np.array_equal(o, r)
            else:
                g1 = bool(np.all(np.abs(c - rc) <= 1e-4 * np.maximum(1.0, np.abs(rc))))
                # fixed-point sums per weight class leave a cell's flow within 2^(e - 19) of the exact sum, 2^e
                # being the largest |flow| in reach (1e-4 up to 64-pixel flows; more for fields that span the frame)
                tol = max(1e-4, 2.0 ** (np.ceil(np.log2(max(1e-9, np.abs(flow).max()))) - 19)) if depth else 1e-4
                g2 = bool(np.all(np.abs(o - r) <= tol * np.maximum(1.0, np.abs(r))))
                g3 = bool(np.array_equal((rc > 0), (c > 0)))
                c2, o2 = torch.empty_like(count), torch.empty_like(out)
                if depth:
                    assert cabi.depthflowprojection_forward(gf, gw, c2, o2, fh) == 0
                else:
                    assert cabi.flowprojection_forward(gf, c2, o2, fh) == 0
                # (a wild field takes the atomic fallback, the reference's own scheme: order-dependent like the reference)
                g4 = kind == "wild" or bool(torch.equal(c2, count) and torch.equal(o2, out))
                good = g1 and g2 and g3 and g4
                if not good:
                    i = np.unravel_index(np.argmax(np.abs(o - r) / np.maximum(1.0, np.abs(r))), o.shape)
                    print("   count close %s, flow close %s, hole mask equal %s, run-to-run equal %s; worst flow cell %s: got %r want %r count %r"
                          % (g1, g2, g3, g4, i, o[i], r[i], rc[i[0], 0, i[2], i[3]]), flush=True)
            if not good:
                ok = False
                print("MISMATCH case %d: B=%d H=%d W=%d %s dyadic=%s fillhole=%d depth=%s max|dcount| %g max|dflow| %g"
                      % (it, B, H, W, kind, dyadic, fh, depth, np.abs(c - rc).max(), np.abs(o - r).max()), flush=True)
    if it % 4 == 0 and kind != "wild":
        # the list form (round 4): this field, a scaled copy and its negative as ONE call == three single calls, bit for bit
        items = [gf, (gf * 0.5).contiguous(), (-gf).contiguous()]
        for depth in (False, True):
            cn = [torch.full((B, 1, H, W), float("nan"), device=dev) for _ in items]
            ou = [torch.full((B, 2, H, W), float("nan"), device=dev) for _ in items]
            assert cabi.flowprojection_forward_batch(items, cn, ou, 1, gw if depth else None) == 0
            for k, fl in enumerate(items):
                c1, o1 = torch.empty_like(cn[k]), torch.empty_like(ou[k])
                if depth:
                    assert cabi.depthflowprojection_forward(fl, gw, c1, o1, 1) == 0
                else:
                    assert cabi.flowprojection_forward(fl, c1, o1, 1) == 0
                if not (torch.equal(c1, cn[k]) and torch.equal(o1, ou[k])):
                    ok = False
                    print("MISMATCH case %d: batched item %d differs from its single call (B=%d H=%d W=%d %s depth=%s)"
                          % (it, k, B, H, W, kind, depth), flush=True)
    bad += 0 if ok else 1
    if it % 25 == 24:
        print("%d cases, %d with mismatches" % (it + 1, bad), flush=True)
print("done: %d cases, %d with mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
