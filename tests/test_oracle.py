"""CPU suite, part 1: pin the oracle.

The reference ships no golden vectors for this path, so the C
restatement (oracle/vfi_oracle.c) is pinned three ways:
  1. against an independent vectorised numpy formulation (oracle/np_oracle.py) --
     bit for bit wherever the op is deterministic;
  2. against the analytic known-answer cases of SURVEY.md section 8(c), items 1-8;
  3. backward passes against properties that do not depend on either implementation
     (linearity / adjoint identities in float64, closed-form gathers);
and then frozen in tests/golden/*.npz, which this file re-checks.
"""
import os

import numpy as np
import pytest

f32 = np.float32


def smooth_flow(rng, b, h, w, sigma):
    from tests.golden.make_golden import smooth_flow as sf
    return sf(rng, b, h, w, sigma)


SHAPES = [(1, 3, 32, 48, 4), (2, 5, 24, 40, 4), (1, 2, 20, 28, 5), (1, 3, 18, 22, 6), (1, 1, 16, 16, 2),
          (1, 4, 7, 9, 4), (1, 2, 5, 70, 3)]


# ------------------------------------------------------------------ 1. C vs numpy

@pytest.mark.parametrize("B,C,H,W,fs", SHAPES)
def test_filterinterp_c_equals_numpy(oracle, np_oracle, B, C, H, W, fs):
    rng = np.random.default_rng(B * 1000 + H * 10 + fs)
    img = rng.random((B, C, H, W), dtype=f32)
    filt = rng.random((B, fs * fs, H, W), dtype=f32)
    for flow in (smooth_flow(rng, B, H, W, 3.0), rng.uniform(-1, 1, (B, 2, H, W)).astype(f32),
                 rng.uniform(-W / 2, W / 2, (B, 2, H, W)).astype(f32)):
        a = oracle.filterinterp_ori_fwd(img, flow, filt)
        assert np.array_equal(a, np_oracle.filterinterp_ori_fwd(img, flow, filt))
        # fused mode differs by rounding only
        assert np.abs(a - oracle.filterinterp_ori_fwd(img, flow, filt, fmad=1)).max() <= 4e-6 * fs * fs
        # threaded run is the same function
        assert np.array_equal(a, oracle.filterinterp_ori_fwd(img, flow, filt, nthreads=3))


@pytest.mark.parametrize("B,C,H,W,fs", SHAPES[:5])
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_deformable_c_equals_numpy(oracle, np_oracle, B, C, H, W, fs, variant):
    rng = np.random.default_rng(77 + variant)
    img = rng.random((B, C, H, W), dtype=f32)
    flow = smooth_flow(rng, B, H, W, 2.0)
    filt = rng.random((B, fs * fs, H, W), dtype=f32)
    for scale in (0.5, 3.0):        # 3.0 pushes displaced taps over the border (clamped corners)
        off = rng.uniform(-scale, scale, (B, 2 * fs * fs, H, W)).astype(f32)
        a = oracle.filterinterp_defor_fwd(variant, img, flow, filt, off)
        assert np.array_equal(a, np_oracle.filterinterp_defor_fwd(variant, img, flow, filt, off))
    if variant == 0 and fs not in (4, 6):
        assert not a.any()          # kernel body exists for fs 4 and 6 only: zeros stay


@pytest.mark.parametrize("B,H,W", [(1, 32, 48), (2, 17, 23), (1, 8, 70)])
def test_projection_c_equals_numpy(oracle, np_oracle, B, H, W):
    rng = np.random.default_rng(H)
    flow = smooth_flow(rng, B, H, W, 3.0)
    depth = rng.uniform(0.1, 1.0, (B, 1, H, W)).astype(f32)
    for fh in (0, 1):
        a, ca = oracle.flowproj_fwd(flow, fh)
        b, cb = np_oracle.flowproj_fwd(flow, fh)
        assert np.array_equal(ca, cb)                       # counts are small integers: exact
        assert np.abs(a - b).max() <= 1e-5                  # float32 running sum vs float64 sum
        # dyadic flow: sums exact in any order -> bit-identical
        fq = (np.round(flow * 8) / 8).astype(f32)
        a, ca = oracle.flowproj_fwd(fq, fh)
        b, cb = np_oracle.flowproj_fwd(fq, fh)
        assert np.array_equal(a, b) and np.array_equal(ca, cb)
        a, ca = oracle.depthflowproj_fwd(flow, depth, fh)
        b, cb = np_oracle.flowproj_fwd(flow, fh, depth)
        assert np.abs(ca - cb).max() <= 1e-5 and np.abs(a - b).max() <= 1e-4


def test_warp_sepconv_c_equals_numpy(oracle, np_oracle):
    rng = np.random.default_rng(5)
    img = rng.random((2, 3, 20, 24), dtype=f32)
    flow = smooth_flow(rng, 2, 20, 24, 3.0)
    assert np.array_equal(oracle.interp_fwd(img, flow), np_oracle.interp_fwd(img, flow))
    for fs in (1, 3, 5, 9):
        v = rng.random((2, fs, 20 - fs + 1, 24 - fs + 1), dtype=f32)
        h = rng.random((2, fs, 20 - fs + 1, 24 - fs + 1), dtype=f32)
        v[0, :, 1, 2] = 0
        assert np.array_equal(oracle.sepconv_fwd(img, v, h), np_oracle.sepconv_fwd(img, v, h))
        a = oracle.sepconvflow_fwd(v, h, 20, 24)
        assert np.array_equal(a, np_oracle.sepconvflow_fwd(v, h))
        assert a[0, 1, 1, 2] == -2000.0


@pytest.mark.parametrize("C,H,W,pad,k,md,s1,s2", [(32, 10, 12, 4, 1, 4, 1, 1), (7, 9, 11, 4, 1, 4, 1, 1),
                                                   (5, 12, 14, 3, 3, 4, 1, 2), (40, 8, 9, 20, 1, 20, 2, 2),
                                                   (70, 6, 7, 4, 1, 4, 1, 1)])
def test_correlation_c_equals_numpy(oracle, np_oracle, C, H, W, pad, k, md, s1, s2):
    rng = np.random.default_rng(C)
    f1 = rng.normal(size=(2, C, H, W)).astype(f32)
    f2 = rng.normal(size=(2, C, H, W)).astype(f32)
    seq = oracle.correlation_fwd(f1, f2, pad, k, md, s1, s2, order=1)
    assert np.array_equal(seq, np_oracle.correlation_fwd(f1, f2, pad, k, md, s1, s2))
    ref64 = np_oracle.correlation_fwd(f1, f2, pad, k, md, s1, s2, dtype=np.float64)
    for order in (0, 1):
        for fmad in (0, 1):
            a = oracle.correlation_fwd(f1, f2, pad, k, md, s1, s2, order=order, fmad=fmad)
            assert np.abs(a - ref64).max() <= 1e-6


# ------------------------------------------------------------------ 2. analytic cases

def test_case1_identity(oracle):
    rng = np.random.default_rng(1)
    img = rng.random((1, 3, 16, 20), dtype=f32)
    flow = np.zeros((1, 2, 16, 20), f32)
    filt = np.zeros((1, 16, 16, 20), f32)
    filt[:, 5] = 1                                           # tap (1,1) = (int(y2), int(x2))
    assert np.array_equal(oracle.filterinterp_ori_fwd(img, flow, filt), img)


def test_case2_integer_shift_and_case3_copy_through(oracle):
    rng = np.random.default_rng(2)
    H, W = 16, 20
    img = rng.random((1, 2, H, W), dtype=f32)
    flow = np.zeros((1, 2, H, W), f32)
    flow[:, 0], flow[:, 1] = 3, -2
    filt = np.zeros((1, 16, H, W), f32)
    filt[:, 5] = 1
    out = oracle.filterinterp_ori_fwd(img, flow, filt)
    ys, xs = np.arange(H) - 2, np.arange(W) + 3
    valid = (ys >= 0)[:, None] & (xs <= W - 1)[None, :]
    shifted = img[:, :, np.clip(ys, 0, H - 1)][:, :, :, np.clip(xs, 0, W - 1)]
    assert np.array_equal(out[0][:, valid], shifted[0][:, valid])
    assert np.array_equal(out[0][:, ~valid], img[0][:, ~valid])     # lands outside -> copy of the input
    flow[:, 0] = W / 2                                       # |fx| >= w/2 -> invalid everywhere
    assert np.array_equal(oracle.filterinterp_ori_fwd(img, flow, filt), img)


def test_case4_bilinear(oracle, np_oracle):
    rng = np.random.default_rng(3)
    H, W = 12, 14
    img = rng.random((1, 1, H, W), dtype=f32)
    flow = np.full((1, 2, H, W), 0.5, f32)
    filt = np.zeros((1, 16, H, W), f32)
    filt[:, [5, 6, 9, 10]] = 1                               # one tap per quadrant, next to the sample point
    out = oracle.filterinterp_ori_fwd(img, flow, filt)
    inner = np_oracle.interp_fwd(img, flow)                  # plain bilinear warp
    assert np.allclose(out[0, 0, :H - 1, :W - 1], inner[0, 0, :H - 1, :W - 1], atol=1e-6)


def test_case5_constant_flow_projection(oracle):
    H, W = 12, 16
    flow = np.zeros((1, 2, H, W), f32)
    flow[:, 0], flow[:, 1] = 2, 1
    out, count = oracle.flowproj_fwd(flow, 0)
    # interior: 2 source columns x 2 source rows reach each target.  The last column / row also
    # receives the R == L (Bm == T) double add of the sources that land exactly on it: 2 + 1 = 3
    assert (count[0, 0, 3:H - 1, 4:W - 1] == 4).all()
    assert (count[0, 0, 3:H - 1, W - 1] == 6).all()
    assert (count[0, 0, H - 1, 4:W - 1] == 6).all()
    assert count[0, 0, H - 1, W - 1] == 9
    assert (count[0, 0, 0, :] == 0).all() and (count[0, 0, :, :2] == 0).all()
    assert (out[0, 0][count[0, 0] > 0] == -2).all() and (out[0, 1][count[0, 0] > 0] == -1).all()
    filled, _ = oracle.flowproj_fwd(flow, 1)
    # holes take the nearest covered value; (0,0) and (0,1) see only holes in all four
    # directions (row 0 and columns 0-1 are uncovered) and stay untouched
    corner = np.zeros((H, W), bool)
    corner[0, :2] = True
    assert (filled[0, 0][~corner] == -2).all() and (filled[0, 1][~corner] == -1).all()
    assert not filled[0][:, corner].any()


def test_case6_all_out_of_frame(oracle):
    flow = np.full((1, 2, 8, 8), 100.0, f32)
    out, count = oracle.flowproj_fwd(flow, 1)
    assert not out.any() and not count.any()


def test_case7_depth_collision(oracle):
    H, W = 6, 8
    flow = np.full((1, 2, H, W), 100.0, f32)                 # everything leaves the frame ...
    depth = np.ones((1, 1, H, W), f32)
    flow[0, :, 2, 2] = (1.0, 0.0)                            # ... except two sources landing on (2,3)
    flow[0, :, 2, 4] = (-1.0, 0.0)
    d1, d2 = f32(0.25), f32(0.75)
    depth[0, 0, 2, 2], depth[0, 0, 2, 4] = d1, d2
    out, count = oracle.depthflowproj_fwd(flow, depth, 0)
    assert count[0, 0, 2, 3] == d1 + d2
    assert np.isclose(out[0, 0, 2, 3], -(d1 * 1.0 + d2 * -1.0) / (d1 + d2))


def test_mindepth_c_equals_numpy_and_analytic(oracle, np_oracle):
    rng = np.random.default_rng(11)
    for (B, H, W, sig) in ((2, 13, 17, 2.0), (1, 24, 40, 4.0), (1, 8, 8, 0.3), (1, 1, 9, 1.0), (1, 7, 1, 1.0)):
        flow = (rng.standard_normal((B, 2, H, W)) * sig).astype(f32)
        wgt = (np.round(rng.uniform(0.1, 1.0, (B, 1, H, W)) * 4) / 4).astype(f32)      # many ties
        for fh in (0, 1):
            a, ca = oracle.mindepthflowproj_fwd(flow, wgt, fh)
            b, cb = np_oracle.mindepthflowproj_fwd(flow, wgt, fh)
            assert np.array_equal(ca, cb) and np.array_equal(a, b)
    # two sources on one target: the nearer one (larger inverse depth) wins, whatever the order
    H, W = 6, 8
    flow = np.full((1, 2, H, W), 100.0, f32)                 # everything else leaves the frame
    wgt = np.ones((1, 1, H, W), f32)
    flow[0, :, 2, 1] = (2.25, 1.5)                           # (1,2) -> (3.25, 3.5) -> target (3,3)
    flow[0, :, 4, 5] = (-1.5, -0.25)                         # (5,4) -> (3.5, 3.75) -> target (3,3)
    for (w0, w1, winner) in ((0.5, 0.75, 1), (0.75, 0.5, 0), (0.5, 0.5, 0)):   # tie: first source in raster order
        wgt[0, 0, 2, 1], wgt[0, 0, 4, 5] = w0, w1
        out, count = oracle.mindepthflowproj_fwd(flow, wgt, 0)
        assert count[0, 0, 3, 3] == max(w0, w1) and (count != 0).sum() == 1
        exp = (-2.25, -1.5) if winner == 0 else (1.5, 0.25)
        assert tuple(out[0, :, 3, 3]) == exp
        filled, _ = oracle.mindepthflowproj_fwd(flow, wgt, 1)
        # holes on row 3 and column 3 copy the only value found; the rest of the frame finds nothing and stays 0
        assert np.all(filled[0, 0, 3, :] == exp[0]) and np.all(filled[0, 1, :, 3] == exp[1])
        assert filled[0, 0, 0, 0] == 0
    # an incoming count above every weight keeps the zeros; non-positive weights never register
    out, count = oracle.mindepthflowproj_fwd(flow, wgt, 0, count0=np.full((1, 1, H, W), 2.0, f32))
    assert not out.any() and np.all(count == 2.0)
    out, count = oracle.mindepthflowproj_fwd(flow, -wgt, 0)
    assert not out.any() and not count.any()
    # backward: a source receives -gout of each of its four neighbours whose count equals its weight
    wgt[0, 0, 2, 1], wgt[0, 0, 4, 5] = 0.5, 0.75
    _, count = oracle.mindepthflowproj_fwd(flow, wgt, 0)
    gout = rng.standard_normal((1, 2, H, W)).astype(f32)
    g = oracle.mindepthflowproj_bwd(flow, wgt, count, gout)
    assert np.array_equal(g[0, :, 4, 5], -gout[0, :, 3, 3]) and not g[0, :, 2, 1].any()
    assert (g != 0).sum() == 2


def test_correlation_half_restatement(oracle):
    """at::Half instantiation: products rounded to half, float accumulation -> close to the float op on the same values."""
    rng = np.random.default_rng(31)
    for (C, H, W, pad, k, md, s1, s2) in ((8, 9, 11, 4, 1, 4, 1, 1), (5, 12, 14, 3, 3, 2, 1, 2), (3, 10, 10, 4, 1, 4, 2, 2)):
        f1 = rng.standard_normal((2, C, H, W)).astype(np.float16)
        f2 = rng.standard_normal((2, C, H, W)).astype(np.float16)
        got = oracle.correlation_fwd_f16(f1, f2, pad, k, md, s1, s2)
        want = oracle.correlation_fwd(f1.astype(f32), f2.astype(f32), pad, k, md, s1, s2)
        assert got.dtype == np.float16 and got.shape == want.shape
        assert np.abs(got.astype(f32) - want).max() <= 4e-3
    # values whose products are exact in half: only the final rounding differs from the float op
    f1 = (rng.integers(-8, 9, (1, 6, 9, 9)) / 4).astype(np.float16)
    f2 = (rng.integers(-8, 9, (1, 6, 9, 9)) / 4).astype(np.float16)
    want = oracle.correlation_fwd(f1.astype(f32), f2.astype(f32)).astype(np.float16)
    assert np.array_equal(oracle.correlation_fwd_f16(f1, f2), want)


def test_case8_correlation_constant(oracle):
    f = np.ones((1, 1, 10, 12), f32)
    out = oracle.correlation_fwd(f, f, 4, 1, 4, 1, 1)
    assert out.shape == (1, 81, 10, 12)
    assert (out[0, 40] == 1).all()                           # centre displacement
    assert (out[0, 0, :4, :] == 0).all() and (out[0, 0, 4:, 4:] == 1).all()   # (-4,-4): zero pad falloff
    assert oracle.correlation_out_dims(10, 12, 20, 1, 20, 2, 2) == (441, 5, 6)


# ------------------------------------------------------------------ 3. backward properties

def _dot(a, b):
    return float(np.sum(a.astype(np.float64) * b.astype(np.float64)))


def test_filterinterp_backward_properties(oracle, np_oracle):
    rng = np.random.default_rng(11)
    B, C, H, W = 1, 3, 14, 18
    img = rng.random((B, C, H, W), dtype=f32)
    flow = smooth_flow(rng, B, H, W, 2.0)
    filt = rng.random((B, 16, H, W), dtype=f32)
    gout = rng.normal(size=(B, C, H, W)).astype(f32)
    gimg, gflow, gfilt = oracle.filterinterp_ori_bwd(img, flow, filt, gout)
    valid = np_oracle._fi_geometry(flow, H, W, 4)[0]
    gv = np.where(valid[:, None], gout, 0).astype(f32)       # invalid pixels get no gradient at all
    # forward is linear in img and in filt: <gout, F(d)> == <grad, d>
    d = rng.normal(size=img.shape).astype(f32)
    zero_copy = np.where(valid[:, None], oracle.filterinterp_ori_fwd(d, flow, filt), 0)
    assert abs(_dot(gv, zero_copy) - _dot(gimg, d)) <= 1e-3
    d = rng.normal(size=filt.shape).astype(f32)
    lin = np.where(valid[:, None], oracle.filterinterp_ori_fwd(img, flow, d), 0)
    assert abs(_dot(gv, lin) - _dot(gfilt, d)) <= 1e-3
    # flow gradient = quadrant differences; quadrant sums via one-hot blends of the numpy forward
    _, _, _, ix, iy, alpha, beta, L, T = np_oracle._fi_geometry(flow, H, W, 4)
    q = []
    for quad in range(4):
        m = np.zeros((B, 16, H, W), f32)
        rows = (0, 1) if quad < 2 else (2, 3)
        cols = (0, 1) if quad % 2 == 0 else (2, 3)
        for r in rows:
            for c in cols:
                m[:, r * 4 + c] = 1
        # sum over the quadrant of img*filt == forward with filt masked and all-ones blend weights
        qsum = np.zeros((B, C, H, W), np.float64)
        for r in rows:
            for c in cols:
                val = np_oracle._gather(img, np.clip(T + r, 0, H - 1), np.clip(L + c, 0, W - 1))
                qsum += val.astype(np.float64) * filt[:, r * 4 + c][:, None]
        q.append(qsum)
    a, b = alpha[:, None].astype(np.float64), beta[:, None].astype(np.float64)
    gx = np.sum(gout * ((1 - b) * (q[1] - q[0]) + b * (q[3] - q[2])), axis=1)
    gy = np.sum(gout * ((1 - a) * (q[2] - q[0]) + a * (q[3] - q[1])), axis=1)
    assert np.abs(np.where(valid, gx, 0) - gflow[:, 0]).max() <= 1e-4
    assert np.abs(np.where(valid, gy, 0) - gflow[:, 1]).max() <= 1e-4


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("fs", [4, 3])
def test_deformable_backward_properties(oracle, np_oracle, variant, fs):
    rng = np.random.default_rng(100 + variant * 10 + fs)
    B, C, H, W = 2, 3, 12, 16
    img = rng.random((B, C, H, W), dtype=f32)
    flow = smooth_flow(rng, B, H, W, 2.0)
    filt = rng.random((B, fs * fs, H, W), dtype=f32)
    off = rng.uniform(-0.9, 0.9, (B, 2 * fs * fs, H, W)).astype(f32)
    gout = rng.normal(size=(B, C, H, W)).astype(f32)
    gimg, gflow, gfilt, goff = oracle.filterinterp_defor_bwd(variant, img, flow, filt, off, gout)
    # 1. independent float64 formulation of all four gradients
    n_gimg, n_gflow, n_gfilt, n_goff = np_oracle.filterinterp_defor_bwd(variant, img, flow, filt, off, gout)
    assert np.abs(gimg - n_gimg).max() <= 2e-4 and np.abs(gflow - n_gflow).max() <= 2e-4
    assert np.abs(goff - n_goff).max() <= 2e-4
    valid = np_oracle._fi_geometry(flow, H, W, fs)[0]
    gv = np.where(valid[:, None], gout, 0).astype(f32)
    if variant == 2:
        assert gfilt is None
    else:
        assert np.abs(gfilt - n_gfilt).max() <= 2e-4
        # 2. the forward is linear in the filter: <gout, F(filt = d)> == <gfilt, d>
        d = rng.normal(size=filt.shape).astype(f32)
        lin = np.where(valid[:, None], oracle.filterinterp_defor_fwd(variant, img, flow, d, off), 0)
        if not (variant == 0 and fs not in (4, 6)):          # that forward has no body for other sizes
            assert abs(_dot(gv, lin) - _dot(gfilt, d)) <= 2e-3
    # 3. the offset gradient is the derivative of the forward where nothing switches
    #    (quadrant membership and integer parts stay put): central differences on a few entries
    if not (variant == 0 and fs not in (4, 6)):
        eps = 1e-3
        checked = 0
        for _ in range(200):
            b, k, y, x = rng.integers(B), rng.integers(2 * fs * fs), rng.integers(2, H - 2), rng.integers(2, W - 2)
            if not valid[b, y, x]:
                continue
            op, om = off.copy(), off.copy()
            op[b, k, y, x] += eps
            om[b, k, y, x] -= eps
            fp = oracle.filterinterp_defor_fwd(variant, img, flow, filt, op)[b, :, y, x].astype(np.float64)
            fm = oracle.filterinterp_defor_fwd(variant, img, flow, filt, om)[b, :, y, x].astype(np.float64)
            f0 = oracle.filterinterp_defor_fwd(variant, img, flow, filt, off)[b, :, y, x].astype(np.float64)
            num = float(np.dot(gout[b, :, y, x], (fp - fm) / (2 * eps)))
            # skip entries where the +-eps steps are not collinear (a tap changed quadrant / cell)
            if np.abs((fp - f0) - (f0 - fm)).max() > 1e-4:
                continue
            assert abs(num - goff[b, k, y, x]) <= 5e-3 * max(1.0, abs(num)), (variant, b, k, y, x, num, goff[b, k, y, x])
            checked += 1
        assert checked >= 20


def test_projection_backward_closed_form(oracle, np_oracle):
    rng = np.random.default_rng(12)
    B, H, W = 2, 12, 16
    flow = smooth_flow(rng, B, H, W, 2.0)
    depth = rng.uniform(0.1, 1.0, (B, 1, H, W)).astype(f32)
    gout = rng.normal(size=(B, 2, H, W)).astype(f32)
    out, count = oracle.depthflowproj_fwd(flow, depth, 0)
    cnt = np.where(count > 0, count, 1).astype(f32)
    valid, fx, fy, L, T, R, Bm = np_oracle._project_targets(flow, H, W)
    exp_plain = np.zeros((B, 2, H, W), np.float64)
    exp_depth = np.zeros((B, 2, H, W), np.float64)
    exp_gd = np.zeros((B, H, W), np.float64)
    bi = np.arange(B)[:, None, None]
    for (ty, tx) in ((T, L), (T, R), (Bm, L), (Bm, R)):
        c = cnt[bi, 0, ty, tx].astype(np.float64)
        for ch, f in ((0, fx), (1, fy)):
            g = gout[bi, ch, ty, tx].astype(np.float64)
            exp_plain[:, ch] += -g / c
            exp_depth[:, ch] += -g * depth[:, 0] / c
            exp_gd += -g / c * (f - out[bi, ch, ty, tx])
    assert np.abs(np.where(valid[:, None], exp_plain, 0) - oracle.flowproj_bwd(flow, cnt, gout)).max() <= 1e-4
    gflow, gdepth = oracle.depthflowproj_bwd(flow, depth, cnt, out, gout)
    assert np.abs(np.where(valid[:, None], exp_depth, 0) - gflow).max() <= 1e-4
    assert np.abs(np.where(valid, exp_gd, 0) - gdepth[:, 0]).max() <= 1e-3


def test_warp_sepconv_backward_properties(oracle):
    rng = np.random.default_rng(13)
    B, C, H, W, fs = 1, 3, 14, 16, 5
    img = rng.random((B, C, H, W), dtype=f32)
    flow = smooth_flow(rng, B, H, W, 2.0)
    gout = rng.normal(size=(B, C, H, W)).astype(f32)
    gimg, gflow = oracle.interp_bwd(img, flow, gout)
    d = rng.normal(size=img.shape).astype(f32)
    assert abs(_dot(gout, oracle.interp_fwd(d, flow)) - _dot(gimg, d)) <= 1e-3
    # the flow gradient is the true derivative of the bilinear warp away from integer crossings
    eps = 1e-3
    num = np.zeros_like(gflow, dtype=np.float64)
    for ch in range(2):
        fp, fm = flow.copy(), flow.copy()
        fp[:, ch] += eps
        fm[:, ch] -= eps
        num[:, ch] = np.sum(gout.astype(np.float64) * (oracle.interp_fwd(img, fp).astype(np.float64) -
                                                       oracle.interp_fwd(img, fm)), axis=1) / (2 * eps)
    frac = (np.arange(W)[None, None, :] + flow[:, 0]) % 1, (np.arange(H)[None, :, None] + flow[:, 1]) % 1
    safe = (np.minimum(frac[0], 1 - frac[0]) > 0.01) & (np.minimum(frac[1], 1 - frac[1]) > 0.01)
    x2 = np.arange(W)[None, None, :] + flow[:, 0]
    y2 = np.arange(H)[None, :, None] + flow[:, 1]
    safe &= (x2 > 0.01) & (y2 > 0.01) & (x2 < W - 1.01) & (y2 < H - 1.01)
    assert np.abs(num - gflow)[:, :, :, :][np.broadcast_to(safe[:, None], num.shape)].max() <= 5e-2
    # separable convolution is linear in each of its three inputs
    v = rng.random((B, fs, H - fs + 1, W - fs + 1), dtype=f32)
    h = rng.random((B, fs, H - fs + 1, W - fs + 1), dtype=f32)
    g = rng.normal(size=(B, C, H - fs + 1, W - fs + 1)).astype(f32)
    gimg, gv, gh = oracle.sepconv_bwd(img, v, h, g)
    d = rng.normal(size=img.shape).astype(f32)
    assert abs(_dot(g, oracle.sepconv_fwd(d, v, h)) - _dot(gimg, d)) <= 2e-3
    d = rng.normal(size=v.shape).astype(f32)
    assert abs(_dot(g, oracle.sepconv_fwd(img, d, h)) - _dot(gv, d)) <= 2e-3
    assert abs(_dot(g, oracle.sepconv_fwd(img, v, d)) - _dot(gh, d)) <= 2e-3
    # SeparableConvFlow: d(m/s)/dk_f = f/s - m/s^2
    gf = rng.normal(size=(B, 2, H - fs + 1, W - fs + 1)).astype(f32)
    gv, gh = oracle.sepconvflow_bwd(v, h, gf, H, W)
    for k, gk, ch in ((v, gv, 1), (h, gh, 0)):
        s = k.sum(1, dtype=np.float64)
        m = sum(f * k[:, f].astype(np.float64) for f in range(fs))
        for f in range(fs):
            assert np.abs(gf[:, ch] * (f / s - m / s ** 2) - gk[:, f]).max() <= 1e-4


def test_correlation_backward_adjoint(oracle):
    rng = np.random.default_rng(14)
    for (C, H, W, pad, k, md, s2) in ((6, 7, 9, 4, 1, 4, 1), (3, 8, 8, 4, 3, 4, 2)):
        f1 = rng.normal(size=(2, C, H, W)).astype(f32)
        f2 = rng.normal(size=(2, C, H, W)).astype(f32)
        out = oracle.correlation_fwd(f1, f2, pad, k, md, 1, s2)
        g = rng.normal(size=out.shape).astype(f32)
        g1, g2 = oracle.correlation_bwd(f1, f2, g, pad, k, md, 1, s2)
        d = rng.normal(size=f1.shape).astype(f32)
        assert abs(_dot(g, oracle.correlation_fwd(d, f2, pad, k, md, 1, s2)) - _dot(g1, d)) <= 1e-3
        assert abs(_dot(g, oracle.correlation_fwd(f1, d, pad, k, md, 1, s2)) - _dot(g2, d)) <= 1e-3


# ------------------------------------------------------------------ 3b. glue either side of the ops
# The reference does these steps with torch built-ins; the oracle restates ATen's kernels and is
# pinned here against the torch CPU build of this image (the dependency itself, importable).

def _torch_warp(feat, flo, align_corners):
    """PWCDCNet.warp written out with torch CPU ops (PWCNet/PWCNet.py:159-199)."""
    import torch
    x, flo = torch.from_numpy(feat), torch.from_numpy(flo)
    B, C, H, W = x.shape
    xx = torch.arange(0, W).view(1, -1).repeat(H, 1).view(1, 1, H, W).repeat(B, 1, 1, 1)
    yy = torch.arange(0, H).view(-1, 1).repeat(1, W).view(1, 1, H, W).repeat(B, 1, 1, 1)
    vgrid = torch.cat((xx, yy), 1).float() + flo
    vgrid[:, 0, :, :] = 2.0 * vgrid[:, 0, :, :].clone() / max(W - 1, 1) - 1.0
    vgrid[:, 1, :, :] = 2.0 * vgrid[:, 1, :, :].clone() / max(H - 1, 1) - 1.0
    vgrid = vgrid.permute(0, 2, 3, 1)
    output = torch.nn.functional.grid_sample(x, vgrid, align_corners=align_corners)
    mask = torch.nn.functional.grid_sample(torch.ones(x.size()), vgrid, align_corners=align_corners)
    mask[mask < 0.9999] = 0
    mask[mask > 0] = 1
    return (output * mask).numpy()


@pytest.mark.parametrize("shape", [(1, 2, 8, 11), (2, 2, 5, 7), (1, 3, 1, 1), (1, 2, 1, 9), (1, 2, 18, 31)])
def test_flow_upsample4_matches_torch(oracle, shape):
    import torch
    rng = np.random.default_rng(sum(shape))
    x = rng.normal(size=shape).astype(f32)
    for m0, m1 in ((20.0, 0.5), (20.0, 0.25), (1.0, 1.0)):
        ref = torch.nn.Upsample(scale_factor=4, mode="bilinear")(torch.from_numpy((m0 * x) * f32(m1))).numpy()
        for fmad in (0, 1):
            got = oracle.flow_upsample4(x, m0, m1, fmad)
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    # a constant stays that constant exactly (1.5 and the weights are dyadic); the first corner is the
    # first input value (source index clamped to 0), the last one is it to rounding (l0*p + l1*p)
    c = oracle.flow_upsample4(np.full(shape, 1.5, f32), 2.0, 0.5)
    assert np.array_equal(c, np.full(c.shape, 1.5, f32))
    got = oracle.flow_upsample4(x, 1.0, 1.0)
    assert np.array_equal(got[:, :, 0, 0], x[:, :, 0, 0])
    assert np.abs(got[:, :, -1, -1] - x[:, :, -1, -1]).max() <= 2.4e-7 * np.abs(x).max()


@pytest.mark.parametrize("align_corners", [True, False])
@pytest.mark.parametrize("shape", [(1, 3, 9, 13), (2, 4, 16, 22), (1, 1, 1, 1), (1, 2, 2, 40)])
def test_pwc_warp_matches_torch(oracle, shape, align_corners):
    rng = np.random.default_rng(sum(shape) + int(align_corners))
    B, C, H, W = shape
    feat = rng.normal(size=shape).astype(f32)
    for scale in (0.7, 3.0, 40.0):
        flo = (rng.normal(size=(B, 2, H, W)) * scale).astype(f32)
        ref = _torch_warp(feat, flo, align_corners)
        for fmad in (0, 1):
            got = oracle.pwc_warp(feat, flo, align_corners, fmad)
            # a sample whose ones-mask sits within rounding of the 0.9999 threshold may flip: none here
            assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(feat).max()), (scale, fmad)
    # everything far outside: zeros (align_corners=True collapses a size-1 axis onto its only pixel)
    if not align_corners or (H > 1 and W > 1):
        assert not oracle.pwc_warp(feat, np.full((B, 2, H, W), 1e4, f32), align_corners).any()
    if align_corners and H > 1 and W > 1:
        # zero flow: the normalisation round-trips to the pixel itself (to rounding)
        got = oracle.pwc_warp(feat, np.zeros((B, 2, H, W), f32), True)
        assert np.abs(got - feat).max() <= 1e-5


def test_frame_boundary_restatement(oracle):
    from vfidkr_amd import fused                      # host logic only: no GPU needed for padding_for
    assert fused.padding_for(480, 640) == (32, 32, 16, 16)          # -> 512 x 704 (demo_MiddleBury.py:317)
    assert fused.padding_for(1080, 1920) == (32, 32, 36, 36)        # -> 1152 x 1984
    assert fused.padding_for(256, 448) == (32, 32, 32, 32)          # -> 320 x 512
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (2, 9, 14, 3), dtype=np.uint8)
    left, right, top, bottom = 3, 2, 4, 1
    x = oracle.frame_to_padded(u8, left, right, top, bottom)
    assert x.dtype == f32 and x.shape == (2, 3, 9 + 5, 14 + 5)
    assert np.array_equal(x[:, :, 0, 0], u8[:, 0, 0, :].astype(f32) / f32(255.0))            # replicated corner
    assert np.array_equal(oracle.padded_to_frame(x, 9, 14, left, top), u8)                    # exact round trip
    # rounding is half to even on 255 * y, clipping first
    y = np.array([-0.2, 0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 0.999, 1.0, 7.0], f32).reshape(1, 1, 1, 8).repeat(3, 1)
    assert oracle.padded_to_frame(y, 1, 8, 0, 0)[0, 0, :, 0].tolist() == [0, 0, 0, 2, 2, 255, 255, 255]
    a = rng.integers(0, 256, (1, 5, 6, 3), dtype=np.uint8)
    err, psnr = oracle.frame_error(a, a)
    assert err == 0.0 and psnr == float("inf")
    b = a.copy()
    b[0, 0, 0, 0] = a[0, 0, 0, 0] ^ 0x10
    err, psnr = oracle.frame_error(a, b)
    assert abs(err - 16.0 / a.size) < 1e-12 and abs(psnr - 20 * np.log10(255.0 / np.sqrt(256.0 / a.size))) < 1e-9


def test_frame_ssim_matches_torch_formulation(oracle):
    """The SSIM the demo reports (demo_MiddleBury.py:40-162, 382-388), written here with torch's conv2d in float32."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(23)
    for (B, h, w) in ((1, 24, 31), (2, 11, 40), (1, 9, 30), (1, 16, 7)):
        gt = rng.integers(0, 256, (B, h, w, 3), dtype=np.uint8)
        rec = np.clip(gt.astype(np.int32) + rng.integers(-12, 13, gt.shape), 0, 255).astype(np.uint8)
        X = torch.from_numpy(rec).permute(0, 3, 1, 2).reshape(B * 3, 1, h, w).float() / 255
        Y = torch.from_numpy(gt).permute(0, 3, 1, 2).reshape(B * 3, 1, h, w).float() / 255
        coords = torch.arange(11, dtype=torch.float32) - 5
        g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
        g = g / g.sum()

        def blur(t):
            if t.shape[2] >= 11:
                t = F.conv2d(t, g.view(1, 1, 11, 1))
            if t.shape[3] >= 11:
                t = F.conv2d(t, g.view(1, 1, 1, 11))
            return t

        mu1, mu2 = blur(X), blur(Y)
        s1, s2, s12 = blur(X * X) - mu1 * mu1, blur(Y * Y) - mu2 * mu2, blur(X * Y) - mu1 * mu2
        cs = (2 * s12 + 0.03 ** 2) / (s1 + s2 + 0.03 ** 2)
        want = float((((2 * mu1 * mu2 + 0.01 ** 2) / (mu1 * mu1 + mu2 * mu2 + 0.01 ** 2)) * cs).mean())
        got = oracle.frame_ssim(rec, gt)
        assert abs(got - want) <= 2e-5, (got, want)
        assert abs(oracle.frame_ssim(gt, gt) - 1.0) <= 1e-12
    flat = np.full((1, 20, 20, 3), 200, np.uint8)
    assert abs(oracle.frame_ssim(flat, flat) - 1.0) <= 1e-12
    # uniform frames of different levels: structure term 1, luminance term (2ab + C1) / (a^2 + b^2 + C1)
    other = np.full((1, 20, 20, 3), 100, np.uint8)
    a, b = np.float64(np.float32(200) / np.float32(255)), np.float64(np.float32(100) / np.float32(255))
    assert abs(oracle.frame_ssim(flat, other) - (2 * a * b + 1e-4) / (a * a + b * b + 1e-4)) <= 1e-9


# ------------------------------------------------------------------ 3b. random shapes (hypothesis): the two formulations agree

def test_random_shapes_c_equals_numpy(oracle, np_oracle):
    """Shapes and flow scales drawn by hypothesis (ragged sizes, 1-pixel dimensions, flows from sub-pixel to beyond
    the frame): the C restatement and the independent numpy formulation agree bit for bit on every deterministic op."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(st.integers(1, 2), st.integers(1, 4), st.integers(1, 19), st.integers(1, 23), st.sampled_from([2, 3, 4, 5]),
           st.sampled_from([0.0, 0.4, 2.5, 40.0]), st.integers(0, 2 ** 31 - 1))
    def run(B, C, H, W, fs, scale, seed):
        rng = np.random.default_rng(seed)
        img = rng.standard_normal((B, C, H, W)).astype(f32)
        filt = rng.random((B, fs * fs, H, W), dtype=f32)
        flow = (rng.standard_normal((B, 2, H, W)) * scale).astype(f32)
        assert np.array_equal(oracle.filterinterp_ori_fwd(img, flow, filt), np_oracle.filterinterp_ori_fwd(img, flow, filt))
        assert np.array_equal(oracle.interp_fwd(img, flow), np_oracle.interp_fwd(img, flow))
        # projection: dyadic flows make every sum exact, so the float64 formulation must agree bit for bit
        fq = (np.round(flow * 8) / 8).astype(f32)
        wgt = (np.round(rng.uniform(0.1, 1.0, (B, 1, H, W)) * 16) / 16 + 1 / 16).astype(f32)
        for fh in (0, 1):
            a, ca = oracle.flowproj_fwd(fq, fh)
            b, cb = np_oracle.flowproj_fwd(fq, fh)
            assert np.array_equal(ca, cb) and np.array_equal(a, b)
            a, ca = oracle.mindepthflowproj_fwd(fq, wgt, fh)
            b, cb = np_oracle.mindepthflowproj_fwd(fq, wgt, fh)
            assert np.array_equal(ca, cb) and np.array_equal(a, b)

    run()


# ------------------------------------------------------------------ 4. golden fixtures still hold

def test_golden_fixtures_reproduced(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "filterinterp.npz"))
    img, flow, filt, off, gout = g["fi_img"], g["fi_flow"], g["fi_filt"], g["fi_off"], g["fi_gout"]
    assert np.array_equal(oracle.filterinterp_ori_fwd(img, flow, filt), g["fi_out"])
    for got, name in zip(oracle.filterinterp_ori_bwd(img, flow, filt, gout), ("fi_gimg", "fi_gflow", "fi_gfilt")):
        assert np.array_equal(got, g[name])
    for v, name in ((0, "offset"), (1, "region"), (2, "nofilter")):
        assert np.array_equal(oracle.filterinterp_defor_fwd(v, img, flow, filt, off), g["fi_out_" + name])
        for got, n in zip(oracle.filterinterp_defor_bwd(v, img, flow, filt, off, gout), ("gimg", "gflow", "gfilt", "goff")):
            assert (got is None and v == 2) or np.array_equal(got, g["fi_%s_%s" % (n, name)])
    assert np.array_equal(oracle.filterinterp_ori_fwd(img[:1], flow[:1], g["fi5_filt"]), g["fi5_out"])
    # the two copy-through pixels planted by the generator
    assert np.array_equal(g["fi_out"][0, :, 5, 7], img[0, :, 5, 7])
    assert np.array_equal(g["fi_out"][1, :, 0, 0], img[1, :, 0, 0])

    g = np.load(os.path.join(golden_dir, "glue.npz"))
    assert np.array_equal(oracle.flow_upsample4(g["flow_q"], 20.0, 0.25), g["up4"])
    for fh in (0, 1):
        out, count = oracle.flowproj_up4_fwd(g["flow_q"], 20.0, 0.25, fh, fmad=0)
        assert np.array_equal(out, g["proj_up4_fh%d" % fh]) and np.array_equal(count, g["proj_up4_count_fh%d" % fh])
    for ac in (0, 1):
        assert np.array_equal(oracle.pwc_warp(g["feat"], g["flo"], bool(ac)), g["warp_ac%d" % ac])
    blend, o0, o2 = oracle.filterinterp_blend(g["ref0"], g["ref2"], g["flow0"], g["flow2"], g["filt0"], g["filt2"],
                                              0.75, 0.25)
    assert np.array_equal(blend, g["blend"]) and np.array_equal(o0, g["blend_out0"]) and np.array_equal(o2, g["blend_out2"])
    assert np.array_equal(oracle.frame_to_padded(g["frame_u8"], 3, 2, 4, 1), g["frame_padded"])
    assert np.array_equal(oracle.padded_to_frame(g["frame_y"], 9, 14, 3, 4), g["frame_back"])

    g = np.load(os.path.join(golden_dir, "projection.npz"))
    for fh in (0, 1):
        out, count = oracle.flowproj_fwd(g["flow"], fh)
        assert np.array_equal(out, g["out_fh%d" % fh]) and np.array_equal(count, g["count_fh%d" % fh])
        out, count = oracle.depthflowproj_fwd(g["flow"], g["depth"], fh)
        assert np.array_equal(out, g["dout_fh%d" % fh]) and np.array_equal(count, g["dcount_fh%d" % fh])

    g = np.load(os.path.join(golden_dir, "mindepth.npz"))
    for fh in (0, 1):
        out, count = oracle.mindepthflowproj_fwd(g["flow"], g["weight"], fh)
        assert np.array_equal(out, g["out_fh%d" % fh]) and np.array_equal(count, g["count_fh%d" % fh])
    assert np.array_equal(oracle.mindepthflowproj_bwd(g["flow"], g["weight"], g["count_fh0"], g["gout"]), g["gflow"])

    g = np.load(os.path.join(golden_dir, "warp_sepconv.npz"))
    assert np.array_equal(oracle.interp_fwd(g["img"], g["flow"]), g["out"])
    assert np.array_equal(oracle.sepconv_fwd(g["img"], g["sep_v"], g["sep_h"]), g["sep_out"])
    assert np.array_equal(oracle.sepconvflow_fwd(g["sep_v"], g["sep_h"], 32, 48), g["sepflow_out"])
    assert (g["sepflow_out"][0, 1, 3, 4] == -2000.0)

    g = np.load(os.path.join(golden_dir, "correlation.npz"))
    assert np.array_equal(oracle.correlation_fwd(g["f1"], g["f2"], 4, 1, 4, 1, 1), g["out_pwc"])
    assert np.array_equal(oracle.correlation_fwd(g["f1"], g["f2"], 3, 3, 4, 1, 2), g["out_k3s2"])
    assert g["out_flownet"].shape == (2, 441, 6, 9)
