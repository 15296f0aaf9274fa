"""Host-side mirrors of the glue either side of the hot-path ops, on the fused entry
points of libvfi_hip.so.  Same names and argument meaning as the reference's static helpers
(`networks/DAIN_slowmotion.py:204-216, 301-335`, `PWCNet/PWCNet.py:159-199`,
`demo_MiddleBury.py:280-318, 350-388`); inference only (no autograd)."""
import math

import torch

from . import cabi


def _check(err, what):
    if err != 0:
        raise RuntimeError("%s: the binding returned %d (shape / stride mismatch)" % (what, err))


class DirectionStreams:
    """One HIP stream per flow direction.  In `DAIN_slowmotion.forward` (networks/DAIN_slowmotion.py:147-183) direction d's
    chain -- its flow network with the correlations, `FlowProject` of its flow for every time offset, `FilterInterpolate_ctx` /
    `FilterInterpolate` on context / frame d -- needs nothing of the other direction's until the blend.  Run on two streams
    the chains overlap: the short launches of one hide under the 196-channel warps of the other (1080p pair: 6.5 instead of
    7.0 ms, same bits).  The library keeps its projection workspace per (device, stream); give each direction its own
    `count` plane and output tensors.

        ds = DirectionStreams(device)
        ds.fork()                       # the side stream starts after what the current stream holds NOW (the inputs)
        for d in (0, 1):
            with ds.direction(d):       # d = 0: the current stream; d = 1: the side stream
                ...                     # direction d's calls
        ds.join()                       # the current stream continues when both are done

    Memory across the two streams (torch's caching allocator keeps one pool per stream): a tensor allocated inside
    `with ds.direction(1)` belongs to the side stream's pool and is consumed on the current stream after `join()`; inputs
    allocated on the current stream are read on the side stream after `fork()`.  That is safe as long as such a tensor stays
    alive until the next `fork()` / `join()` has ordered the streams again -- a tensor freed earlier can be handed out again
    on the OTHER stream while launches that use it are still in flight.  Either pre-allocate what crosses the streams (what
    `bench.py` does), or call `tensor.record_stream(stream)` on it; `join(*tensors)` does the latter for the tensors it is given.
    """

    def __init__(self, device=None):
        self.device = device
        self.side = torch.cuda.Stream(device)
        self._forked = False

    def fork(self):
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        self._forked = True

    def direction(self, d):
        if not self._forked:
            raise RuntimeError("DirectionStreams.fork() first: the side stream must be ordered after the inputs")
        return torch.cuda.stream(torch.cuda.current_stream(self.device) if d == 0 else self.side)

    def join(self, *tensors):
        """`tensors`: results of direction 1 that the current stream goes on to use (recorded on it for the allocator)"""
        if self._forked:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.side)
            for t in tensors:
                t.record_stream(cur)
            self._forked = False


def forward_flownets_upsample(flow_q, div_flow, time_offsets):
    """`forward_flownets` after the flow network: [div_flow * flow * t upsampled x4 for t in time_offsets]."""
    b, c, hq, wq = flow_q.shape
    outs = []
    for t in time_offsets:
        out = torch.empty((b, c, 4 * hq, 4 * wq), device=flow_q.device, dtype=torch.float32)
        _check(cabi.flow_upsample4(flow_q, out, float(div_flow), float(t)), "flow_upsample4")
        outs.append(out)
    return outs


def FlowProject(inputs, depth=None, fillhole=True):
    """`DAIN.FlowProject` / `DAIN_slowmotion.FlowProject` (networks/DAIN.py:533-539, networks/DAIN_slowmotion.py:301-307):
    inputs = the list of full-resolution flows of `forward_flownets`, depth = the direction's inverse depth (or None);
    returns the list of projected flows.  The reference loops over the list, one module call per flow; here the list is
    ONE call of the library (one launch triple per eight flows), same results bit for bit.  depth may also be a list, one
    tensor per flow -- which is how both directions go through together: `FlowProject_directions`."""
    inputs = list(inputs)
    counts = [torch.empty((f.size(0), 1, f.size(2), f.size(3)), device=f.device, dtype=torch.float32) for f in inputs]
    outs = [torch.empty_strided(f.shape, f.stride(), device=f.device, dtype=torch.float32) for f in inputs]
    _check(cabi.flowprojection_forward_batch(inputs, counts, outs, int(fillhole), depth), "flowprojection_forward_batch")
    return outs


def FlowProject_directions(cur_offset_outputs, depth_inv=None, fillhole=True):
    """The two `FlowProject` calls of `forward` back to back (networks/DAIN.py:215-220, networks/DAIN_slowmotion.py:156-159):
    `[FlowProject(cur_offset_outputs[0], depth_inv[0]), FlowProject(cur_offset_outputs[1], depth_inv[1])]` as one call."""
    n0 = len(cur_offset_outputs[0])
    flows = list(cur_offset_outputs[0]) + list(cur_offset_outputs[1])
    depth = None if depth_inv is None else [depth_inv[0]] * n0 + [depth_inv[1]] * len(cur_offset_outputs[1])
    outs = FlowProject(flows, depth, fillhole)
    return [outs[:n0], outs[n0:]]


def FlowProject_from_quarter(flow_q, div_flow, time_offsets, depth=None, fillhole=True):
    """`forward_flownets` + `FlowProject` (inference: fillhole) in one call per time offset: the full-resolution flow lives in
    a scratch tensor of the library, not in a tensor of the caller."""
    b, _, hq, wq = flow_q.shape
    outs = []
    for t in time_offsets:
        count = torch.empty((b, 1, 4 * hq, 4 * wq), device=flow_q.device, dtype=torch.float32)
        out = torch.empty((b, 2, 4 * hq, 4 * wq), device=flow_q.device, dtype=torch.float32)
        if depth is None:
            err = cabi.flowprojection_forward_up4(flow_q, count, out, float(div_flow), float(t), int(fillhole))
        else:
            err = cabi.depthflowprojection_forward_up4(flow_q, depth, count, out, float(div_flow), float(t), int(fillhole))
        _check(err, "flowprojection_forward_up4")
        outs.append(out)
    return outs


def FilterInterpolate(ref0, ref2, offset, filter, filter_size2, time_offset):
    """`DAIN.FilterInterpolate`: returns (ref0_offset*(1-t) + ref2_offset*t, ref0_offset, ref2_offset)."""
    assert filter[0].size(1) == filter_size2
    blend, out0, out2 = torch.empty_like(ref0), torch.empty_like(ref0), torch.empty_like(ref0)
    _check(cabi.filterinterp_blend_forward(ref0, ref2, offset[0], offset[1], filter[0], filter[1], blend, out0, out2,
                                           float(1.0 - time_offset), float(time_offset)), "filterinterp_blend_forward")
    return blend, out0, out2


def FilterInterpolate_ctx_all(ctx0, ctx2, offsets, filter):
    """`DAIN_slowmotion.FilterInterpolate_ctx` (networks/DAIN_slowmotion.py:311-317) for every time offset of a step at
    once (the loop at :167-183 calls it once per t with the same context tensors and filters): offsets[d][t] is the
    projected flow of direction d at time offset t.  Returns [(ctx0_offset_t, ctx2_offset_t) for t], each pair what
    the reference's call returns -- from one launch per direction that stages every image window once."""
    nt = len(offsets[0])
    # (outputs share the input's layout, strided views included: empty_like would densify a channel slice)
    out0 = [torch.empty_strided(ctx0.shape, ctx0.stride(), dtype=ctx0.dtype, device=ctx0.device) for _ in range(nt)]
    out2 = [torch.empty_strided(ctx2.shape, ctx2.stride(), dtype=ctx2.dtype, device=ctx2.device) for _ in range(nt)]
    _check(cabi.filterinterp_forward_ori_multi(ctx0, list(offsets[0]), filter[0], out0), "filterinterp_forward_ori_multi")
    _check(cabi.filterinterp_forward_ori_multi(ctx2, list(offsets[1]), filter[1], out2), "filterinterp_forward_ori_multi")
    return list(zip(out0, out2))


def warp(x, flo, align_corners=True):
    """`PWCDCNet.warp`."""
    out = torch.empty_like(x)
    _check(cabi.pwc_warp_forward(x, flo, out, align_corners), "pwc_warp_forward")
    return out


def warp_corr(c1, c2, flo, align_corners=True, one_launch=False):
    """`self.corr(c1, self.warp(c2, flo))` of PWCDCNet.forward (PWCNet/PWCNet.py:244-247 ...); the reference applies
    its LeakyReLU to the result afterwards.  Default: the warp kernel, then the correlation kernel.  one_launch=True:
    vfi_pwc_warp_correlation_forward, which never materialises the warped tensor -- same bits, but measured SLOWER on
    MI355X at PWC-Net's sizes (0.33 vs 0.15 ms for the five levels of a 1080p pair): every tile re-forms the bilinear
    samples of its 4-pixel halo (3.75 x the taps), which costs more than the 2 x 18 MB round trip it saves."""
    if one_launch:
        return cabi.pwc_warp_correlation_forward(c1, c2, flo, align_corners)
    return cabi.correlation_forward(c1, warp(c2, flo, align_corners), 4, 1, 4, 1, 1)


def corr_pair(c1_a, c2_a, c1_b, c2_b):
    """`self.corr(c1, c2)` of the two flow networks of a frame pair at one pyramid level (PWCNet/PWCNet.py:230, 246, 267, 283,
    300 for the (I0, I1) network and again for (I1, I0): networks/DAIN.py:196-202) in one launch; returns both cost volumes."""
    return cabi.correlation_forward_pair(c1_a, c2_a, c1_b, c2_b, 4, 1, 4, 1, 1)


def padding_for(height, width):
    """(left, right, top, bottom) of `demo_MiddleBury.py:294-310`: next multiple of 128, or 32 each side."""
    def one(n):
        if n != ((n >> 7) << 7):
            total = (((n >> 7) + 1) << 7) - n
            return int(total / 2), total - int(total / 2)
        return 32, 32
    left, right = one(width)
    top, bottom = one(height)
    return left, right, top, bottom


def frames_to_padded(frames_u8):
    """uint8 [B,h,w,3] on the GPU -> float32 [B,3,H,W] / 255 with replication padding; returns (tensor, padding)."""
    b, h, w, _ = frames_u8.shape
    left, right, top, bottom = padding_for(h, w)
    out = torch.empty((b, 3, h + top + bottom, w + left + right), device=frames_u8.device, dtype=torch.float32)
    _check(cabi.frame_u8_to_planar(frames_u8, out, left, right, top, bottom), "frame_u8_to_planar")
    return out, (left, right, top, bottom)


def padded_to_frames(y, height, width, padding):
    """float32 [B,3,H,W] -> uint8 [B,height,width,3]: clip, crop, x255, round (`demo_MiddleBury.py:350-364`)."""
    left, _, top, _ = padding
    out = torch.empty((y.size(0), height, width, 3), device=y.device, dtype=torch.uint8)
    _check(cabi.planar_to_frame_u8(y, out, top, left), "planar_to_frame_u8")
    return out


def interpolation_error_and_psnr(rec_u8, gt_u8):
    """(mean |rec - gt|, PSNR in dB) of `demo_MiddleBury.py:370-381`; the sums are exact integers."""
    sums = torch.zeros(2, device=rec_u8.device, dtype=torch.int64)
    _check(cabi.frame_error_sums(rec_u8, gt_u8, sums), "frame_error_sums")
    s_abs, s_sq = (int(v) for v in sums.cpu())
    n = rec_u8.numel()
    mse = s_sq / n
    psnr = float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
    return s_abs / n, psnr


def ssim(rec_u8, gt_u8):
    """Mean SSIM of uint8 frames [B,h,w,3] as `demo_MiddleBury.py:382-388` computes it (`ssim()` :40-162 on the
    colour planes / 255: 11-tap sigma-1.5 Gaussian without padding, data_range 1)."""
    sums = torch.zeros(1, device=rec_u8.device, dtype=torch.int64)
    _check(cabi.frame_ssim_sums(rec_u8, gt_u8, sums), "frame_ssim_sums")
    b, h, w, _ = rec_u8.shape
    n = b * 3 * (h - 10 if h >= 11 else h) * (w - 10 if w >= 11 else w)
    return int(sums.cpu()[0]) / 4294967296.0 / n
