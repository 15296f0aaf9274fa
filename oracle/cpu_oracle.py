"""ctypes binding of oracle/libvfi_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY (see vfi_oracle.h).  Every function takes and returns
dense float32 NCHW numpy arrays and mirrors one reference op:

  filterinterp_ori_fwd/bwd   filterinterpolation_cuda_kernel.cu:2692-3125
  filterinterp_defor_fwd     filterinterpolation_cuda_kernel.cu:29-426, 1353-1496, 2070-2191
  flowproj_fwd/bwd           flowprojection_cuda_kernel.cu:29-301
  depthflowproj_fwd/bwd      depthflowprojection_cuda_kernel.cu:29-341
  mindepthflowproj_fwd/bwd   mindepthflowprojection_cuda_kernel.cu:27-331 (sequential raster order)
  interp_fwd/bwd             interpolation_cuda_kernel.cu:29-202
  sepconv_fwd/bwd            separableconv_cuda_kernel.cu:29-135
  sepconvflow_fwd/bwd        separableconvflow_cuda_kernel.cu:29-173
  correlation_fwd/bwd        correlation_cuda_kernel.cu:47-334
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvfi_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.c_int


def build(force=False):
    """Compile libvfi_oracle.so with the committed Makefile (gcc)."""
    src = os.path.join(_HERE, "vfi_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def set_num_threads(n):
    """Threads for the OpenMP loops of filterinterp_ori_fwd (nthreads < 1) and correlation_fwd."""
    lib().vfi_oracle_set_num_threads(int(n))


def _p(a):
    return a.ctypes.data_as(_F)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _check(err, what):
    if err != 0:
        raise RuntimeError("oracle %s returned %d" % (what, err))


def filterinterp_ori_fwd(img, flow, filt, fmad=0, nthreads=1):
    img, flow, filt = _f32(img), _f32(flow), _f32(filt)
    B, C, H, W = img.shape
    assert flow.shape == (B, 2, H, W) and filt.shape[0] == B and filt.shape[2:] == (H, W)
    out = np.zeros_like(img)
    _check(lib().vfi_oracle_filterinterp_ori_fwd(_p(img), _p(flow), _p(filt), _p(out), B, C, H, W,
                                                 filt.shape[1], int(fmad), int(nthreads)), "filterinterp_ori_fwd")
    return out


def filterinterp_ori_bwd(img, flow, filt, gout, fmad=0):
    img, flow, filt, gout = _f32(img), _f32(flow), _f32(filt), _f32(gout)
    B, C, H, W = img.shape
    gimg, gflow, gfilt = np.zeros_like(img), np.zeros_like(flow), np.zeros_like(filt)
    _check(lib().vfi_oracle_filterinterp_ori_bwd(_p(img), _p(flow), _p(filt), _p(gout), _p(gimg), _p(gflow),
                                                 _p(gfilt), B, C, H, W, filt.shape[1], int(fmad)),
           "filterinterp_ori_bwd")
    return gimg, gflow, gfilt


def filterinterp_defor_fwd(variant, img, flow, filt, off, fmad=0):
    """variant 0: 4-input forward; 1: deforconv; 2: nofilterwithdeforconv (filt ignored)."""
    img, flow, off = _f32(img), _f32(flow), _f32(off)
    B, C, H, W = img.shape
    fs = int(np.sqrt(np.float32(off.shape[1] // 2)))
    if variant != 2:
        filt = _f32(filt)
        fs = int(np.sqrt(np.float32(filt.shape[1])))
        fp = _p(filt)
    else:
        fp = None
    out = np.zeros_like(img)
    _check(lib().vfi_oracle_filterinterp_defor_fwd(int(variant), _p(img), _p(flow), fp, _p(off), _p(out),
                                                   B, C, H, W, fs, int(fmad)), "filterinterp_defor_fwd")
    return out


def filterinterp_defor_bwd(variant, img, flow, filt, off, gout, fmad=0):
    """Returns (gimg, gflow, gfilt or None, goff)."""
    img, flow, off, gout = _f32(img), _f32(flow), _f32(off), _f32(gout)
    B, C, H, W = img.shape
    gimg, gflow, goff = np.zeros_like(img), np.zeros_like(flow), np.zeros_like(off)
    if variant != 2:
        filt = _f32(filt)
        fs = int(np.sqrt(np.float32(filt.shape[1])))
        gfilt = np.zeros_like(filt)
        fp, gfp = _p(filt), _p(gfilt)
    else:
        fs = int(np.sqrt(np.float32(off.shape[1] // 2)))
        gfilt, fp, gfp = None, None, None
    _check(lib().vfi_oracle_filterinterp_defor_bwd(int(variant), _p(img), _p(flow), fp, _p(off), _p(gout), _p(gimg),
                                                   _p(gflow), gfp, _p(goff), B, C, H, W, fs, int(fmad)),
           "filterinterp_defor_bwd")
    return gimg, gflow, gfilt, goff


def flowproj_fwd(flow, fillhole=1):
    flow = _f32(flow)
    B, _, H, W = flow.shape
    count = np.zeros((B, 1, H, W), np.float32)
    out = np.zeros_like(flow)
    _check(lib().vfi_oracle_flowproj_fwd(_p(flow), _p(count), _p(out), B, H, W, int(fillhole)), "flowproj_fwd")
    return out, count


def flowproj_bwd(flow, count, gout):
    flow, count, gout = _f32(flow), _f32(count), _f32(gout)
    B, _, H, W = flow.shape
    gflow = np.zeros_like(flow)
    _check(lib().vfi_oracle_flowproj_bwd(_p(flow), _p(count), _p(gout), _p(gflow), B, H, W), "flowproj_bwd")
    return gflow


def depthflowproj_fwd(flow, depth, fillhole=1):
    flow, depth = _f32(flow), _f32(depth)
    B, _, H, W = flow.shape
    count = np.zeros((B, 1, H, W), np.float32)
    out = np.zeros_like(flow)
    _check(lib().vfi_oracle_depthflowproj_fwd(_p(flow), _p(depth), _p(count), _p(out), B, H, W, int(fillhole), 0),
           "depthflowproj_fwd")
    return out, count


def depthflowproj_bwd(flow, depth, count, out, gout):
    flow, depth, count, out, gout = _f32(flow), _f32(depth), _f32(count), _f32(out), _f32(gout)
    B, _, H, W = flow.shape
    gflow, gdepth = np.zeros_like(flow), np.zeros_like(depth)
    _check(lib().vfi_oracle_depthflowproj_bwd(_p(flow), _p(depth), _p(count), _p(out), _p(gout), _p(gflow),
                                              _p(gdepth), B, H, W), "depthflowproj_bwd")
    return gflow, gdepth


def mindepthflowproj_fwd(flow, weight, fillhole=1, count0=None):
    """count0: optional incoming `count` (the reference wrapper passes zeros); out starts at zero."""
    flow, weight = _f32(flow), _f32(weight)
    B, _, H, W = flow.shape
    count = np.zeros((B, 1, H, W), np.float32) if count0 is None else _f32(count0).copy()
    out = np.zeros_like(flow)
    _check(lib().vfi_oracle_mindepthflowproj_fwd(_p(flow), _p(weight), _p(count), _p(out), B, H, W, int(fillhole)),
           "mindepthflowproj_fwd")
    return out, count


def mindepthflowproj_bwd(flow, weight, count, gout):
    flow, weight, count, gout = _f32(flow), _f32(weight), _f32(count), _f32(gout)
    B, _, H, W = flow.shape
    gflow = np.zeros_like(flow)
    _check(lib().vfi_oracle_mindepthflowproj_bwd(_p(flow), _p(weight), _p(count), _p(gout), _p(gflow), B, H, W),
           "mindepthflowproj_bwd")
    return gflow


def interp_fwd(img, flow, fmad=0):
    img, flow = _f32(img), _f32(flow)
    B, C, H, W = img.shape
    out = np.zeros_like(img)
    _check(lib().vfi_oracle_interp_fwd(_p(img), _p(flow), _p(out), B, C, H, W, int(fmad)), "interp_fwd")
    return out


def interp_bwd(img, flow, gout, fmad=0):
    img, flow, gout = _f32(img), _f32(flow), _f32(gout)
    B, C, H, W = img.shape
    gimg, gflow = np.zeros_like(img), np.zeros_like(flow)
    _check(lib().vfi_oracle_interp_bwd(_p(img), _p(flow), _p(gout), _p(gimg), _p(gflow), B, C, H, W, int(fmad)),
           "interp_bwd")
    return gimg, gflow


def sepconv_fwd(img, v, h, fmad=0):
    img, v, h = _f32(img), _f32(v), _f32(h)
    B, C, H, W = img.shape
    fs = v.shape[1]
    out = np.zeros((B, C, H - fs + 1, W - fs + 1), np.float32)
    _check(lib().vfi_oracle_sepconv_fwd(_p(img), _p(v), _p(h), _p(out), B, C, H, W, fs, int(fmad)), "sepconv_fwd")
    return out


def sepconv_bwd(img, v, h, gout):
    img, v, h, gout = _f32(img), _f32(v), _f32(h), _f32(gout)
    B, C, H, W = img.shape
    fs = v.shape[1]
    gimg, gv, gh = np.zeros_like(img), np.zeros_like(v), np.zeros_like(h)
    _check(lib().vfi_oracle_sepconv_bwd(_p(img), _p(v), _p(h), _p(gout), _p(gimg), _p(gv), _p(gh), B, C, H, W, fs),
           "sepconv_bwd")
    return gimg, gv, gh


def sepconvflow_fwd(v, h, H, W, fmad=0):
    v, h = _f32(v), _f32(h)
    B, fs = v.shape[0], v.shape[1]
    out = np.zeros((B, 2, H - fs + 1, W - fs + 1), np.float32)
    _check(lib().vfi_oracle_sepconvflow_fwd(_p(v), _p(h), _p(out), B, H, W, fs, int(fmad)), "sepconvflow_fwd")
    return out


def sepconvflow_bwd(v, h, gflow, H, W, fmad=0):
    v, h, gflow = _f32(v), _f32(h), _f32(gflow)
    B, fs = v.shape[0], v.shape[1]
    gv, gh = np.zeros_like(v), np.zeros_like(h)
    _check(lib().vfi_oracle_sepconvflow_bwd(_p(v), _p(h), _p(gflow), _p(gv), _p(gh), B, H, W, fs, int(fmad)),
           "sepconvflow_bwd")
    return gv, gh


def correlation_out_dims(H, W, pad, k, md, s1, s2):
    oc, oh, ow = _I(), _I(), _I()
    _check(lib().vfi_oracle_correlation_out_dims(H, W, pad, k, md, s1, s2, ctypes.byref(oc), ctypes.byref(oh),
                                                 ctypes.byref(ow)), "correlation_out_dims")
    return oc.value, oh.value, ow.value


def correlation_fwd(f1, f2, pad=4, k=1, md=4, s1=1, s2=1, order=0, fmad=0):
    f1, f2 = _f32(f1), _f32(f2)
    B, C, H, W = f1.shape
    oc, oh, ow = correlation_out_dims(H, W, pad, k, md, s1, s2)
    out = np.zeros((B, oc, oh, ow), np.float32)
    _check(lib().vfi_oracle_correlation_fwd(_p(f1), _p(f2), _p(out), B, C, H, W, pad, k, md, s1, s2, int(order),
                                            int(fmad)), "correlation_fwd")
    return out


def correlation_fwd_f16(f1, f2, pad=4, k=1, md=4, s1=1, s2=1):
    """The at::Half instantiation of the forward (correlation_cuda_kernel.cu:80-146 with scalar_t = Half): every
    product rounded to half (:124), float accumulation in channel order, the mean rounded to half (:143).
    numpy, vectorised over pixels; small inputs."""
    assert f1.dtype == np.float16 and f2.dtype == np.float16
    B, C, H, W = f1.shape
    oc, oh, ow = correlation_out_dims(H, W, pad, k, md, s1, s2)
    kr, dr = (k - 1) // 2, md // s2
    dsz = 2 * dr + 1
    p1 = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.float32)
    p2 = np.zeros_like(p1)
    p1[:, :, pad:pad + H, pad:pad + W] = f1
    p2[:, :, pad:pad + H, pad:pad + W] = f2
    out = np.zeros((B, oc, oh, ow), np.float16)
    ys = np.arange(oh) * s1 + md
    xs = np.arange(ow) * s1 + md
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            acc = np.zeros((B, oh, ow), np.float32)
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    a = p1[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    b = p2[:, :, (ys + tj * s2 + j)[:, None], (xs + ti * s2 + i)[None, :]]
                    prod = (a * b).astype(np.float16).astype(np.float32)     # exact product, one rounding to half
                    for c in range(C):
                        acc = (acc + prod[:, c]).astype(np.float32)
            out[:, (tj + dr) * dsz + (ti + dr)] = (acc / np.float32(k * k * C)).astype(np.float16)
    return out


def correlation_bwd(f1, f2, gout, pad=4, k=1, md=4, s1=1, s2=1):
    f1, f2, gout = _f32(f1), _f32(f2), _f32(gout)
    B, C, H, W = f1.shape
    g1, g2 = np.zeros_like(f1), np.zeros_like(f2)
    _check(lib().vfi_oracle_correlation_bwd(_p(f1), _p(f2), _p(gout), _p(g1), _p(g2), B, C, H, W, pad, k, md, s1,
                                            s2), "correlation_bwd")
    return g1, g2


# ---------------------------------------------------------------- glue either side of the ops

def flow_upsample4(x, m0, m1, fmad=0):
    x = _f32(x)
    B, C, hq, wq = x.shape
    out = np.zeros((B, C, 4 * hq, 4 * wq), np.float32)
    _check(lib().vfi_oracle_flow_upsample4(_p(x), _p(out), B, C, hq, wq, ctypes.c_float(m0), ctypes.c_float(m1),
                                           int(fmad)), "flow_upsample4")
    return out


def flowproj_up4_fwd(flow_q, m0, m1, fillhole=1, depth=None, fmad=1):
    """forward_flownets + FlowProject: the composition the fused GPU entry points compute."""
    flow = flow_upsample4(flow_q, m0, m1, fmad)
    return flowproj_fwd(flow, fillhole) if depth is None else depthflowproj_fwd(flow, depth, fillhole)


def filterinterp_blend(ref0, ref2, flow0, flow2, filt0, filt2, w0, w2, fmad=0):
    """DAIN.FilterInterpolate: (out0 * w0 + out2 * w2, out0, out2), float32 throughout."""
    out0 = filterinterp_ori_fwd(ref0, flow0, filt0, fmad)
    out2 = filterinterp_ori_fwd(ref2, flow2, filt2, fmad)
    return out0 * np.float32(w0) + out2 * np.float32(w2), out0, out2


def pwc_warp(x, flo, align_corners=True, fmad=0):
    x, flo = _f32(x), _f32(flo)
    B, C, H, W = x.shape
    out = np.zeros_like(x)
    _check(lib().vfi_oracle_pwc_warp(_p(x), _p(flo), _p(out), B, C, H, W, int(bool(align_corners)), int(fmad)),
           "pwc_warp")
    return out


def frame_to_padded(frames_u8, left, right, top, bottom):
    """demo_MiddleBury.py:280-318: transpose, astype(float32) / 255.0, ReplicationPad2d."""
    x = np.transpose(frames_u8, (0, 3, 1, 2)).astype("float32") / 255.0
    return np.pad(x, ((0, 0), (0, 0), (top, bottom), (left, right)), mode="edge")


def padded_to_frame(y, height, width, left, top):
    """demo_MiddleBury.py:350-364: 255.0 * y.clip(0, 1.0) cropped, transposed, np.round, uint8."""
    v = np.transpose(255.0 * y.clip(0, 1.0)[:, :, top:top + height, left:left + width], (0, 2, 3, 1))
    return np.round(v).astype(np.uint8)


def frame_error(rec_u8, gt_u8):
    """demo_MiddleBury.py:370-381: (mean |diff|, PSNR)."""
    import math
    diff = 128.0 + rec_u8 - gt_u8
    err = float(np.mean(np.abs(diff - 128.0)))
    mse = float(np.mean((diff - 128.0) ** 2))
    return err, (float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse)))


def frame_ssim(rec_u8, gt_u8):
    """demo_MiddleBury.py:382-388 -> ssim() :40-162 with its defaults: every colour plane / 255 is a single-channel
    image; 11-tap Gaussian (sigma 1.5) along H then W, no padding (a dimension below 11 is not smoothed, :112-119);
    data_range 1, K = (0.01, 0.03); mean of the SSIM map.  float64 throughout (the reference runs it in float32).
    rec_u8, gt_u8: [B,h,w,3] uint8."""
    x = np.moveaxis(rec_u8.astype(np.float32) / np.float32(255), 3, 1).astype(np.float64)      # ToTensor
    y = np.moveaxis(gt_u8.astype(np.float32) / np.float32(255), 3, 1).astype(np.float64)
    coords = np.arange(11, dtype=np.float64) - 11 // 2
    g = np.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()

    def blur(t):
        for axis in (2, 3):
            n = t.shape[axis]
            if n < 11:
                continue
            t = sum(g[k] * np.take(t, np.arange(k, n - 10 + k), axis=axis) for k in range(11))
        return t

    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mu1, mu2 = blur(x), blur(y)
    s1 = blur(x * x) - mu1 * mu1
    s2 = blur(y * y) - mu2 * mu2
    s12 = blur(x * y) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    return float((((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs).mean())


def filterinterp_ori_fwd_f16(img16, flow, filt, fmad=1, nthreads=1):
    """fp16 storage, fp32 arithmetic: the fp32 op on the widened image, rounded to half once."""
    assert img16.dtype == np.float16
    return filterinterp_ori_fwd(img16.astype(np.float32), flow, filt, fmad, nthreads).astype(np.float16)
