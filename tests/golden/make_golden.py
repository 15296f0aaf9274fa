"""Generates tests/golden/*.npz: small seeded inputs + expected outputs of every op.

The reference holds no golden vectors and cannot run here, so the
expected outputs come from the CPU restatement (oracle/vfi_oracle.c, strict mode,
fmad=0), after tests/test_oracle.py has cross-checked it against the independent
numpy formulation and the analytic cases.  The fixtures freeze that behaviour:
the CPU suite checks the oracle still reproduces them, the GPU suite compares
the HIP kernels with them without needing the oracle at all.

    python tests/golden/make_golden.py          # rewrites the fixtures
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cpu_oracle as o  # noqa: E402


def smooth_flow(rng, b, h, w, sigma):
    """Bilinear x4 upsample of quarter-resolution noise (numpy only, deterministic)."""
    lo = rng.normal(0, sigma, (b, 2, h // 4 + 2, w // 4 + 2)).astype(np.float32)
    ys = (np.arange(h) + 0.5) / 4
    xs = (np.arange(w) + 0.5) / 4
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    wy = (ys - y0).astype(np.float32)[None, None, :, None]
    wx = (xs - x0).astype(np.float32)[None, None, None, :]
    a = lo[:, :, y0][:, :, :, x0]
    b_ = lo[:, :, y0][:, :, :, x0 + 1]
    c = lo[:, :, y0 + 1][:, :, :, x0]
    d = lo[:, :, y0 + 1][:, :, :, x0 + 1]
    return ((1 - wy) * ((1 - wx) * a + wx * b_) + wy * ((1 - wx) * c + wx * d)).astype(np.float32)


def main():
    rng = np.random.default_rng(20250202)
    B, C, H, W = 2, 3, 32, 48
    img = rng.random((B, C, H, W), dtype=np.float32)
    flow = smooth_flow(rng, B, H, W, 3.0)
    flow[0, 0, 5, 7] = W            # |fx| >= w/2 -> copy-through
    flow[1, 1, 0, 0] = -3.0         # lands outside -> copy-through
    filt = rng.random((B, 16, H, W), dtype=np.float32)
    off = rng.uniform(-1, 1, (B, 32, H, W)).astype(np.float32)
    gout = rng.normal(size=(B, C, H, W)).astype(np.float32)
    depth = rng.uniform(0.1, 1.0, (B, 1, H, W)).astype(np.float32)

    g = {}
    g["fi_img"], g["fi_flow"], g["fi_filt"], g["fi_off"], g["fi_gout"] = img, flow, filt, off, gout
    g["fi_out"] = o.filterinterp_ori_fwd(img, flow, filt)
    g["fi_gimg"], g["fi_gflow"], g["fi_gfilt"] = o.filterinterp_ori_bwd(img, flow, filt, gout)
    for v, name in ((0, "offset"), (1, "region"), (2, "nofilter")):
        g["fi_out_" + name] = o.filterinterp_defor_fwd(v, img, flow, filt, off)
        d = o.filterinterp_defor_bwd(v, img, flow, filt, off, gout)
        for arr, n in zip(d, ("gimg", "gflow", "gfilt", "goff")):
            if arr is not None:
                g["fi_%s_%s" % (n, name)] = arr
    filt5 = rng.random((1, 25, H, W), dtype=np.float32)
    g["fi5_filt"] = filt5
    g["fi5_out"] = o.filterinterp_ori_fwd(img[:1], flow[:1], filt5)
    np.savez_compressed(os.path.join(HERE, "filterinterp.npz"), **g)

    g = {"flow": flow, "depth": depth}
    # dyadic flow: every partial sum is exact, so any accumulation order gives the same bits
    g["flow_q"] = (np.round(flow * 8) / 8).astype(np.float32)
    for fh in (0, 1):
        g["out_fh%d" % fh], g["count_fh%d" % fh] = o.flowproj_fwd(flow, fh)
        g["outq_fh%d" % fh], g["countq_fh%d" % fh] = o.flowproj_fwd(g["flow_q"], fh)
        g["dout_fh%d" % fh], g["dcount_fh%d" % fh] = o.depthflowproj_fwd(flow, depth, fh)
    gproj = rng.normal(size=(B, 2, H, W)).astype(np.float32)
    g["gout"] = gproj
    g["gflow"] = o.flowproj_bwd(flow, np.where(g["count_fh0"] > 0, g["count_fh0"], 1).astype(np.float32), gproj)
    cnt = np.where(g["dcount_fh0"] > 0, g["dcount_fh0"], 1).astype(np.float32)
    g["dgflow"], g["dgdepth"] = o.depthflowproj_bwd(flow, depth, cnt, g["dout_fh0"], gproj)
    np.savez_compressed(os.path.join(HERE, "projection.npz"), **g)

    g = {"img": img, "flow": flow, "gout": gout}
    g["out"] = o.interp_fwd(img, flow)
    g["gimg"], g["gflow"] = o.interp_bwd(img, flow, gout)
    fs = 5
    v = rng.random((B, fs, H - fs + 1, W - fs + 1), dtype=np.float32)
    h = rng.random((B, fs, H - fs + 1, W - fs + 1), dtype=np.float32)
    v[0, :, 3, 4] = 0.0             # zero weight sum -> -2000 sentinel
    g["sep_v"], g["sep_h"] = v, h
    g["sep_out"] = o.sepconv_fwd(img, v, h)
    gsep = rng.normal(size=g["sep_out"].shape).astype(np.float32)
    g["sep_gout"] = gsep
    g["sep_gimg"], g["sep_gv"], g["sep_gh"] = o.sepconv_bwd(img, v, h, gsep)
    g["sepflow_out"] = o.sepconvflow_fwd(v, h, H, W)
    gsf = rng.normal(size=g["sepflow_out"].shape).astype(np.float32)
    g["sepflow_gout"] = gsf
    g["sepflow_gv"], g["sepflow_gh"] = o.sepconvflow_bwd(v, h, gsf, H, W)
    np.savez_compressed(os.path.join(HERE, "warp_sepconv.npz"), **g)

    g = {}
    f1 = rng.normal(size=(2, 40, 12, 18)).astype(np.float32)
    f2 = rng.normal(size=(2, 40, 12, 18)).astype(np.float32)
    g["f1"], g["f2"] = f1, f2
    g["out_pwc"] = o.correlation_fwd(f1, f2, 4, 1, 4, 1, 1, order=0)
    gc = rng.normal(size=g["out_pwc"].shape).astype(np.float32)
    g["gout_pwc"] = gc
    g["g1_pwc"], g["g2_pwc"] = o.correlation_bwd(f1, f2, gc, 4, 1, 4, 1, 1)
    g["out_k3s2"] = o.correlation_fwd(f1, f2, 3, 3, 4, 1, 2, order=0)
    g["out_flownet"] = o.correlation_fwd(f1[:, :8], f2[:, :8], 20, 1, 20, 2, 2, order=0)
    np.savez_compressed(os.path.join(HERE, "correlation.npz"), **g)

    # glue either side of the ops; its own generator, so the fixtures above keep their values
    rng = np.random.default_rng(20250203)
    g = {}
    g["flow_q"] = (rng.normal(size=(2, 2, 6, 9)) * 0.1).astype(np.float32)       # x 20 x t: a few pixels
    g["up4"] = o.flow_upsample4(g["flow_q"], 20.0, 0.25)
    for fh in (0, 1):
        g["proj_up4_fh%d" % fh], g["proj_up4_count_fh%d" % fh] = o.flowproj_up4_fwd(g["flow_q"], 20.0, 0.25, fh, fmad=0)
    g["feat"] = rng.normal(size=(2, 4, 12, 17)).astype(np.float32)
    g["flo"] = (rng.normal(size=(2, 2, 12, 17)) * 3).astype(np.float32)
    for ac in (0, 1):
        g["warp_ac%d" % ac] = o.pwc_warp(g["feat"], g["flo"], bool(ac))
    g["ref0"], g["ref2"] = img[:1], img[1:]
    g["flow0"], g["flow2"] = flow[:1], flow[1:]
    g["filt0"], g["filt2"] = filt[:1], filt[1:]
    g["blend"], g["blend_out0"], g["blend_out2"] = o.filterinterp_blend(g["ref0"], g["ref2"], g["flow0"], g["flow2"],
                                                                         g["filt0"], g["filt2"], 0.75, 0.25)
    g["frame_u8"] = rng.integers(0, 256, (2, 9, 14, 3), dtype=np.uint8)
    g["frame_padded"] = o.frame_to_padded(g["frame_u8"], 3, 2, 4, 1)
    g["frame_y"] = (g["frame_padded"] * 1.3 - 0.1).astype(np.float32)           # values below 0 and above 1 too
    g["frame_back"] = o.padded_to_frame(g["frame_y"], 9, 14, 3, 4)
    np.savez_compressed(os.path.join(HERE, "glue.npz"), **g)
    # MinDepthFlowProjection: defined result = sequential raster order; weights with ties
    rng = np.random.default_rng(20250204)
    g = {}
    g["flow"] = (rng.normal(size=(2, 2, 19, 27)) * 2.5).astype(np.float32)
    g["weight"] = (np.round(rng.uniform(0.1, 1.0, (2, 1, 19, 27)) * 8) / 8).astype(np.float32)
    for fh in (0, 1):
        g["out_fh%d" % fh], g["count_fh%d" % fh] = o.mindepthflowproj_fwd(g["flow"], g["weight"], fh)
    g["gout"] = rng.normal(size=(2, 2, 19, 27)).astype(np.float32)
    g["gflow"] = o.mindepthflowproj_bwd(g["flow"], g["weight"], g["count_fh0"], g["gout"])
    np.savez_compressed(os.path.join(HERE, "mindepth.npz"), **g)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
