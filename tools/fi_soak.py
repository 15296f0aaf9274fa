#!/usr/bin/env python3
"""Soak: the staged FilterInterpolation kernels (fp32 lean loop with 4- / 8-byte tap reads chosen per tile, the multi-flow
kernel) against the direct-gather kernel on many random frames -- sizes around tile edges, channel counts around the ring
depth and the channel-group split, flows from smooth to rough to leaving the frame, strided views.  Bitwise equality.
    python tools/fi_soak.py [cases] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vfidkr_amd  # noqa: E402,F401
from vfidkr_amd import cabi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260930)
dev = torch.device("cuda:0")
gen = torch.Generator(device="cpu").manual_seed(int(rng.integers(1 << 31)))
bad = 0
for it in range(cases):
    B = int(rng.choice([1, 1, 2, 3]))
    C = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 13, 24, 40, 67]))
    H = int(rng.choice([1, 7, 16, 17, 31, 48, 80, 127, 200, 300]))
    W = int(rng.choice([1, 5, 63, 64, 65, 128, 190, 333, 512, 700]))
    img = torch.randn(B, C, H, W, generator=gen).to(dev)
    filt = torch.rand(B, 16, H, W, generator=gen).to(dev)
    kind = rng.choice(["smooth", "rough", "mixed", "wild", "tiny"])
    base = torch.randn(B, 2, max(1, H // 16 + 1), max(1, W // 16 + 1), generator=gen) * float(rng.choice([1.0, 4.0, 10.0]))
    flow = (torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear", align_corners=True) if H > 1 and W > 1
            else torch.randn(B, 2, H, W, generator=gen) * 2.0)
    if kind == "rough":
        flow = flow + torch.randn(B, 2, H, W, generator=gen) * float(rng.choice([2.0, 6.0, 12.0]))
    elif kind == "mixed":
        flow[:, :, :, W // 3:2 * W // 3] += torch.randn(B, 2, H, 2 * W // 3 - W // 3, generator=gen) * 8.0
    elif kind == "wild":
        flow = (torch.rand(B, 2, H, W, generator=gen) - 0.5) * float(W)
    elif kind == "tiny":
        flow = flow * 0.01
    flow = flow.contiguous().to(dev)
    sliced = rng.random() < 0.3 and C > 1               # channel slices of wider tensors: non-dense batch strides
    def buf(fill=None):                                 # (outputs carry input1's strides: filterinterpolation_cuda.cc:579-583)
        t = torch.empty(B, C + 3 if sliced else C, H, W, device=dev)
        if fill is not None:
            t.fill_(fill)
        return t[:, 2:2 + C] if sliced else t
    if sliced:
        wide = torch.randn(B, C + 3, H, W, generator=gen).to(dev)
        img = wide[:, 2:2 + C]
    ref = buf()
    out = buf(float("nan"))
    assert cabi.filterinterp_forward_ori(img, flow, filt, ref, direct=True) == 0
    assert cabi.filterinterp_forward_ori(img, flow, filt, out) == 0
    same = lambda a, b: torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))      # noqa: E731
    ok = same(out, ref)
    nfl = int(rng.choice([2, 3]))
    flows = [(flow * s).contiguous() for s in (0.5, 1.0, 1.5)[:nfl]]
    outs = [buf(float("nan")) for _ in range(nfl)]
    assert cabi.filterinterp_forward_ori_multi(img, flows, filt, outs) == 0
    for f, o in zip(flows, outs):
        assert cabi.filterinterp_forward_ori(img, f, filt, ref, direct=True) == 0
        ok = ok and same(o, ref)
    # other filter sizes (staged fs = 2, 5, 6 against the direct kernel) and the three deformable variants (staged against
    # the general kernel), on a dense copy, every few cases
    if it % 3 == 0 and H * W <= 120000:
        imgd = torch.empty(B, C, H, W, device=dev).copy_(img)
        for fs in (2, 5, 6):
            fl = torch.rand(B, fs * fs, H, W, generator=gen).to(dev)
            a, b_ = torch.full_like(imgd, float("nan")), torch.empty_like(imgd)
            assert cabi.filterinterp_forward_ori(imgd, flow, fl, a) == 0
            assert cabi.filterinterp_forward_ori(imgd, flow, fl, b_, direct=True) == 0
            ok = ok and same(a, b_)
        off = (torch.randn(B, 32, H, W, generator=gen) * float(rng.choice([0.3, 1.5, 6.0]))).to(dev)
        for variant in (0, 1, 2):
            a, b_ = torch.full_like(imgd, float("nan")), torch.empty_like(imgd)
            i3, i4 = (off, None) if variant == 2 else (filt, off)
            assert cabi.filterinterp_forward_defor(variant, imgd, flow, i3, i4, a) == 0
            assert cabi.filterinterp_forward_defor(variant, imgd, flow, i3, i4, b_, general=True) == 0
            ok = ok and same(a, b_)
    # fp16 storage: the staged kernel against the direct one (folded weights: within the fp16 tolerance of the tests), and the
    # fused blend (second launch's epilogue) against two launches and torch's three elementwise ops, bit for bit
    if it % 2 == 0:
        imgd = torch.empty(B, C, H, W, device=dev).copy_(img)      # (dense: `contiguous()` keeps a B = 1 slice's batch stride)
        i16 = imgd.half()
        a16, b16 = torch.zeros_like(i16), torch.zeros_like(i16)
        assert cabi.filterinterp_forward_ori_f16(i16, flow, filt, a16) == 0
        assert cabi.filterinterp_forward_ori_f16(i16, flow, filt, b16, direct=True) == 0
        ok = ok and bool(((a16.float() - b16.float()).abs() <= 2e-3 * b16.float().abs().clamp(min=1.0)).all())
        if C <= 4:
            img2 = torch.randn(B, C, H, W, generator=gen).to(dev)
            flow2 = (flow * -0.7).contiguous()
            filt2 = torch.rand(B, 16, H, W, generator=gen).to(dev)
            o0, o2, bl = (torch.full_like(imgd, float("nan")) for _ in range(3))
            rc = cabi.filterinterp_blend_forward(imgd, img2, flow, flow2, filt, filt2, bl, o0, o2, 0.75, 0.25)
            assert rc == 0, "blend rc %r: B=%d C=%d H=%d W=%d %s" % (rc, B, C, H, W, kind)
            r0, r2 = torch.empty_like(imgd), torch.empty_like(imgd)
            assert cabi.filterinterp_forward_ori(imgd, flow, filt, r0, direct=True) == 0
            assert cabi.filterinterp_forward_ori(img2, flow2, filt2, r2, direct=True) == 0
            ok = ok and same(o0, r0) and same(o2, r2) and same(bl, r0 * 0.75 + r2 * 0.25)
    if not ok:
        bad += 1
        print("MISMATCH case %d: B=%d C=%d H=%d W=%d %s" % (it, B, C, H, W, kind), flush=True)
    if it % 50 == 49:
        print("%d cases, %d mismatches" % (it + 1, bad), flush=True)
print("done: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
