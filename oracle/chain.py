"""The oracle-only call chain of one interpolated frame -- TEST INFRASTRUCTURE like everything under oracle/.

`FlowProject -> FilterInterpolate -> blend -> crop / x255 / round` as `DAIN_slowmotion.forward` runs it for one time offset
(networks/DAIN_slowmotion.py:156-183, 301-335; frame boundary demo_MiddleBury.py:350-364), every op by the CPU
restatement, nothing shared with the GPU side: the harness-level parity check (PSNR(build, oracle) >= 60 dB
fp32, >= 45 dB fp16 storage; PSNR as demo_MiddleBury.py:370-378) feeds both chains the same inputs and compares the uint8
frames -- unlike the per-op tests, which hand one side's projected flow to the other so that the warp compare is exact.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

from . import cpu_oracle as oracle


def unit(frames, flows, depths, filters, t, height, width, left, top, nthreads=1, ctx=None):
    """frames / flows / depths / filters: [direction 0, direction 1] of padded float32 arrays ([1,3,H,W], [1,2,H,W] = the
    full-resolution flow of `forward_flownets` for time offset t, [1,1,H,W], [1,16,H,W]); ctx: optional pair of context
    tensors [1,C,H,W] (FilterInterpolate_ctx).  Returns dict(proj=[p0, p2], blend, u8, ctx=[c0, c2] or None)."""
    proj = [oracle.depthflowproj_fwd(flows[d], depths[d], 1)[0] for d in range(2)]
    out0 = oracle.filterinterp_ori_fwd(frames[0], proj[0], filters[0], fmad=1, nthreads=nthreads)
    out2 = oracle.filterinterp_ori_fwd(frames[1], proj[1], filters[1], fmad=1, nthreads=nthreads)
    blend = out0 * np.float32(1.0 - t) + out2 * np.float32(t)          # (networks/DAIN_slowmotion.py:335)
    u8 = oracle.padded_to_frame(blend, height, width, left, top)
    warped = None
    if ctx is not None:
        warped = [oracle.filterinterp_ori_fwd(ctx[d], proj[d], filters[d], fmad=1, nthreads=nthreads) for d in range(2)]
    return {"proj": proj, "blend": blend, "u8": u8, "ctx": warped}


def psnr_u8(a, b):
    """demo_MiddleBury.py:370-378 on uint8 frames; 99.0 when identical."""
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return 99.0 if mse == 0 else float(20.0 * np.log10(255.0 / np.sqrt(mse)))


def int_flips(proj_a, proj_b):
    """Pixels whose FilterInterpolation window origin int(x + fx), int(y + fy) differs between two projected flows
    (filterinterpolation_cuda_kernel.cu:2737-2738): where a last-bit difference of the projection becomes a whole-pixel
    difference of the warp."""
    _, _, h, w = proj_a.shape
    xs = np.arange(w, dtype=np.float32)[None, :]
    ys = np.arange(h, dtype=np.float32)[:, None]
    n = 0
    for b in range(proj_a.shape[0]):
        ax, ay = np.trunc(xs + proj_a[b, 0]), np.trunc(ys + proj_a[b, 1])
        bx, by = np.trunc(xs + proj_b[b, 0]), np.trunc(ys + proj_b[b, 1])
        n += int(np.count_nonzero((ax != bx) | (ay != by)))
    return n
