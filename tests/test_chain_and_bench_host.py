"""CPU tests (round 4): the oracle-only frame chain used by the harness-level parity check, and the host-side arithmetic of
bench.py's new blocks."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

f32 = np.float32


def test_chain_unit_is_the_composition_of_the_ops(oracle):
    """oracle/chain.py::unit = DepthFlowProjection of both directions -> FilterInterpolation of both frames -> blend -> crop,
    x255, round (networks/DAIN_slowmotion.py:156-183, 324-335; demo_MiddleBury.py:350-364), nothing else."""
    from oracle import chain
    rng = np.random.default_rng(5)
    h, w, top, left = 20, 36, 6, 4
    H, W = h + 2 * top, w + 2 * left
    frames = [rng.random((1, 3, H, W)).astype(f32) for _ in range(2)]
    flows = [rng.normal(0, 2.0, (1, 2, H, W)).astype(f32) for _ in range(2)]
    depths = [rng.uniform(0.1, 1.0, (1, 1, H, W)).astype(f32) for _ in range(2)]
    filters = [rng.random((1, 16, H, W)).astype(f32) for _ in range(2)]
    ctx = [rng.normal(size=(1, 5, H, W)).astype(f32) for _ in range(2)]
    t = 0.25
    got = chain.unit(frames, flows, depths, filters, t, h, w, left, top, ctx=ctx)
    p = [oracle.depthflowproj_fwd(flows[d], depths[d], 1)[0] for d in range(2)]
    o = [oracle.filterinterp_ori_fwd(frames[d], p[d], filters[d], fmad=1) for d in range(2)]
    blend = o[0] * f32(1.0 - t) + o[1] * f32(t)
    u8 = np.round(np.transpose(255.0 * blend.clip(0, 1.0)[:, :, top:top + h, left:left + w], (0, 2, 3, 1))).astype(np.uint8)
    assert np.array_equal(got["proj"][0], p[0]) and np.array_equal(got["proj"][1], p[1])
    assert np.array_equal(got["blend"], blend) and np.array_equal(got["u8"], u8)
    assert got["u8"].shape == (1, h, w, 3)
    for d in range(2):
        assert np.array_equal(got["ctx"][d], oracle.filterinterp_ori_fwd(ctx[d], p[d], filters[d], fmad=1))


def test_chain_psnr_and_flip_count():
    from oracle import chain
    a = np.zeros((1, 4, 4, 3), np.uint8)
    b = a.copy()
    assert chain.psnr_u8(a, b) == 99.0
    b[0, 0, 0, 0] = 16                                       # mse = 256 / 48
    assert abs(chain.psnr_u8(a, b) - 20 * math.log10(255.0 / math.sqrt(256.0 / 48.0))) < 1e-9
    # int(x + fx): a flow that ends just below an integer against one just above it
    pa = np.zeros((1, 2, 3, 5), f32)
    pb = pa.copy()
    pa[0, 0, 1, 2] = f32(0.99999)                            # x = 2: 2.99999 -> 2
    pb[0, 0, 1, 2] = f32(1.00001)                            #        3.00001 -> 3
    pb[0, 1, 2, 4] = f32(0.5)                                # y = 2: 2.5 -> 2, no flip
    assert chain.int_flips(pa, pb) == 1


def test_best_schedule_block_is_the_median_of_the_repetitions():
    import bench
    shared = {"_two_streams_reps_ms": [6.0, 5.0, 5.5, 5.25, 5.75], "other": 1}
    blk = bench.best_schedule_block(shared, 400.0, 10)
    assert "_two_streams_reps_ms" not in shared
    assert blk["ms_per_step"] == 5.5 and blk["frames_per_s"] == round(3 / 5.5e-3, 1)
    assert blk["spread"]["min"] == round(3 / 6.0e-3, 1) and blk["spread"]["max"] == round(3 / 5.0e-3, 1)
    assert blk["vs_value"] == round((3 / 5.5e-3) / 400.0, 3)
