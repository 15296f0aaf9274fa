"""ctypes caller of libvfi_hip.so (include/vfi_hip.h) for torch GPU tensors.

Each function is the ctypes twin of one reference binding: it applies the
binding's own shape/stride checks (returning 1 silently on a mismatch, like
`filterinterpolation_cuda.cc:543-583`), then passes raw device pointers, sizes,
element strides and torch's current HIP stream to the C ABI.  No arithmetic
happens on this side.  torch is used for memory and streams only.
"""
import ctypes
import math

import torch  # must be imported before libvfi_hip.so so both bind the same libamdhip64.so.7

from . import LIB_PATH

_i = ctypes.c_int
_f = ctypes.c_float
_p = ctypes.c_void_p


class Strides(ctypes.Structure):
    _fields_ = [("b", ctypes.c_int64), ("c", ctypes.c_int64), ("h", ctypes.c_int64)]


VFI_OK, VFI_ERR_SHAPE, VFI_ERR_LAUNCH = 0, 1, -2
DEFOR_OFFSET, DEFOR_REGION, DEFOR_NOFILTER = 0, 1, 2

# name -> argtypes, exactly the declarations of include/vfi_hip.h
SIGNATURES = {
    "vfi_filterinterp_forward_ori": [_p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_filterinterp_backward_ori": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_filterinterp_forward_ori_f16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_filterinterp_forward_ori_multi": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_filterinterp_forward_defor": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides,
                                       Strides, _p],
    "vfi_filterinterp_backward_defor": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides,
                                        Strides, Strides, _p],
    "vfi_flowprojection_forward": [_p, _p, _p, _i, _i, _i, _i, Strides, Strides, _p],
    "vfi_flowprojection_forward_batch": [_p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, _p],
    "vfi_depthflowprojection_forward_batch": [_p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_flowprojection_backward": [_p, _p, _p, _p, _i, _i, _i, Strides, Strides, _p],
    "vfi_projection_reserve": [_i, _i, _i, _p],
    "vfi_release_workspaces": [],
    "vfi_depthflowprojection_forward": [_p, _p, _p, _p, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_depthflowprojection_backward": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_mindepthflowprojection_forward": [_p, _p, _p, _p, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_mindepthflowprojection_backward": [_p, _p, _p, _p, _p, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_interpolation_forward": [_p, _p, _p, _i, _i, _i, _i, Strides, Strides, _p],
    "vfi_interpolation_backward": [_p, _p, _p, _p, _p, _i, _i, _i, _i, Strides, Strides, _p],
    "vfi_separableconv_forward": [_p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides, Strides, _p],
    "vfi_separableconv_backward": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides,
                                   Strides, _p],
    "vfi_separableconvflow_forward": [_p, _p, _p, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_separableconvflow_backward": [_p, _p, _p, _p, _p, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_correlation_output_dims": [_i, _i, _i, _i, _i, _i, _i, ctypes.POINTER(_i), ctypes.POINTER(_i),
                                    ctypes.POINTER(_i)],
    "vfi_correlation_forward": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "vfi_correlation_forward_pair": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "vfi_correlation_forward_f16": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "vfi_correlation_backward": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    # glue either side of the ops
    "vfi_flow_upsample4": [_p, _p, _i, _i, _i, _i, _f, _f, Strides, Strides, _p],
    "vfi_flowprojection_forward_up4": [_p, _p, _p, _i, _i, _i, _f, _f, _i, Strides, Strides, Strides, _p],
    "vfi_depthflowprojection_forward_up4": [_p, _p, _p, _p, _i, _i, _i, _f, _f, _i, Strides, Strides, Strides, Strides,
                                            _p],
    "vfi_filterinterp_blend_forward": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, Strides, Strides,
                                       Strides, Strides, _p],
    "vfi_pwc_warp_forward": [_p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_pwc_warp_correlation_forward": [_p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, _p],
    "vfi_frame_u8_to_planar": [_p, _p, _i, _i, _i, _i, _i, _i, _i, Strides, _p],
    "vfi_planar_to_frame_u8": [_p, _p, _i, _i, _i, _i, _i, Strides, _p],
    "vfi_frame_error_sums": [_p, _p, ctypes.c_int64, _p, _p],
    "vfi_frame_ssim_sums": [_p, _p, _i, _i, _i, _p, _p],
}
# internal entry points used by the bench / tests to time one code path in isolation
INTERNAL_SIGNATURES = {
    "vfi_filterinterp_forward_ori_direct": [_p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_filterinterp_forward_ori_f16_direct": [_p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides, _p],
    "vfi_filterinterp_forward_defor_general": [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, Strides, Strides, Strides,
                                               Strides, _p],
}

_lib = None


def lib():
    """Load libvfi_hip.so; raises OSError if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        l = ctypes.CDLL(LIB_PATH)
        l.vfi_version.restype = ctypes.c_char_p
        l.vfi_version.argtypes = []
        for table in (SIGNATURES, INTERNAL_SIGNATURES):
            for name, argtypes in table.items():
                fn = getattr(l, name)
                fn.restype = _i
                fn.argtypes = argtypes
        _lib = l
    return _lib


def version():
    return lib().vfi_version().decode()


def _st(t):
    return Strides(t.stride(0), t.stride(1), t.stride(2))


def _st_tuple(t):
    return (t.stride(0), t.stride(1), t.stride(2), t.stride(3))


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _dev(t, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("vfidkr_amd.cabi: tensors must live on the GPU (there is no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError("vfidkr_amd.cabi: tensors must be %s" % str(dtype).replace("torch.", ""))
    return t.device


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(_dev(t)).cuda_stream)


def _finish(err):
    if err == VFI_ERR_LAUNCH:
        raise RuntimeError("CUDA call failed")      # the reference's AT_ERROR text
    return err


# ---------------------------------------------------------------- filterinterpolation_cuda

def _fi_checks(input1, input2, input3, out_like):
    b, c, h, w = input1.shape
    if input2.size(0) != b or input2.size(1) != 2 or input2.size(2) != h or input2.size(3) != w:
        return None
    if input1.stride(3) != 1 or input2.stride(3) != 1 or input3.stride(3) != 1:
        return None
    if out_like is not None and (input1.stride(0) != out_like.stride(0) or input1.stride(1) != out_like.stride(1)):
        return None
    return b, c, h, w


def filterinterp_forward_ori(input1, input2, input3, output, direct=False):
    dims = _fi_checks(input1, input2, input3, output)
    if dims is None:
        return 1
    b, c, h, w = dims
    with torch.cuda.device(_dev(input1)):
        fn = lib().vfi_filterinterp_forward_ori_direct if direct else lib().vfi_filterinterp_forward_ori
        return _finish(fn(_ptr(input1), _ptr(input2), _ptr(input3), _ptr(output), b, c, h, w, input3.size(1),
                          _st(input1), _st(input2), _st(input3), _stream(input1)))


def filterinterp_forward_ori_multi(input1, flows, input3, outputs):
    """outputs[t] = FilterInterpolation(input1, flows[t], input3): the time offsets of a slow-motion step in one launch."""
    n = len(flows)
    if n == 0 or len(outputs) != n:
        return 1
    for fl, out in zip(flows, outputs):
        # (layouts compared as _same_strides does: the stride of a size-1 dimension is arbitrary, e.g. after slicing)
        if _fi_checks(input1, fl, input3, out) is None or not _same_strides(fl, flows[0]) or not _same_strides(out, input1):
            return 1
        _dev(fl), _dev(out)
    b, c, h, w = input1.shape
    fp = (ctypes.c_void_p * n)(*[fl.data_ptr() for fl in flows])
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outputs])
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_filterinterp_forward_ori_multi(_ptr(input1), fp, _ptr(input3), op, n, b, c, h, w, input3.size(1),
                                                                _st(input1), _st(flows[0]), _st(input3), _stream(input1)))


def filterinterp_forward_ori_f16(input1, input2, input3, output, direct=False):
    """fp16 storage: input1 / output are float16 tensors, flow and filter float32."""
    dims = _fi_checks(input1, input2, input3, output)
    if dims is None or output.stride(3) != 1:
        return 1
    b, c, h, w = dims
    _dev(input2), _dev(input3), _dev(output, torch.float16)
    with torch.cuda.device(_dev(input1, torch.float16)):
        fn = lib().vfi_filterinterp_forward_ori_f16_direct if direct else lib().vfi_filterinterp_forward_ori_f16
        return _finish(fn(_ptr(input1), _ptr(input2), _ptr(input3), _ptr(output), b, c, h, w, input3.size(1),
                          _st(input1), _st(input2), _st(input3), _stream(input2)))


def filterinterp_backward_ori(input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3):
    dims = _fi_checks(input1, input2, input3, gradinput1)
    if dims is None:
        return 1
    if input2.stride(0) != gradinput2.stride(0) or input2.stride(1) != gradinput2.stride(1):
        return 1
    if input3.stride(1) != gradinput3.stride(1):
        return 1
    b, c, h, w = dims
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_filterinterp_backward_ori(
            _ptr(input1), _ptr(input2), _ptr(input3), _ptr(gradoutput), _ptr(gradinput1), _ptr(gradinput2),
            _ptr(gradinput3), b, c, h, w, input3.size(1), _st(input1), _st(input2), _st(input3), _stream(input1)))


def filterinterp_forward_defor(variant, input1, input2, input3, input4, output, general=False):
    """variant: DEFOR_OFFSET (4-input forward), DEFOR_REGION (deforconv), DEFOR_NOFILTER (input4 = None).
    general=True: always the one-thread-per-pixel kernel (an internal entry point the tests compare the staged one with)."""
    dims = _fi_checks(input1, input2, input3, output if variant == DEFOR_NOFILTER else None)
    if dims is None:
        return 1
    b, c, h, w = dims
    if variant == DEFOR_NOFILTER:
        fs = int(math.sqrt(input3.size(1) // 2))
        p4, s4 = ctypes.c_void_p(0), _st(input3)
    else:
        if input4.stride(3) != 1:
            return 1
        fs = int(math.sqrt(input3.size(1)))
        p4, s4 = _ptr(input4), _st(input4)
    with torch.cuda.device(_dev(input1)):
        fn = lib().vfi_filterinterp_forward_defor_general if general else lib().vfi_filterinterp_forward_defor
        return _finish(fn(variant, _ptr(input1), _ptr(input2), _ptr(input3), p4, _ptr(output), b, c, h, w, fs, _st(input1),
                          _st(input2), _st(input3), s4, _stream(input1)))


def filterinterp_backward_defor(variant, input1, input2, input3, input4, gradoutput, gradinput1, gradinput2,
                                gradinput3, gradinput4):
    """Backward of the deformable variants; variant DEFOR_NOFILTER: input4 = gradinput4 = None."""
    dims = _fi_checks(input1, input2, input3, gradinput1)
    if dims is None:
        return 1
    if input2.stride(0) != gradinput2.stride(0) or input2.stride(1) != gradinput2.stride(1):
        return 1
    if input3.stride(1) != gradinput3.stride(1):
        return 1
    b, c, h, w = dims
    if variant == DEFOR_NOFILTER:
        fs = int(math.sqrt(input3.size(1) // 2))
        p4, g4, s4 = ctypes.c_void_p(0), ctypes.c_void_p(0), _st(input3)
    else:
        if input4.stride(3) != 1:
            return 1
        fs = int(math.sqrt(input3.size(1)))
        p4, g4, s4 = _ptr(input4), _ptr(gradinput4), _st(input4)
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_filterinterp_backward_defor(
            variant, _ptr(input1), _ptr(input2), _ptr(input3), p4, _ptr(gradoutput), _ptr(gradinput1),
            _ptr(gradinput2), _ptr(gradinput3), g4, b, c, h, w, fs, _st(input1), _st(input2), _st(input3), s4,
            _stream(input1)))


# ---------------------------------------------------------------- flowprojection_cuda / depthflowprojection_cuda

def flowprojection_forward(input1, count, output, fillhole):
    if input1.size(1) != 2:
        return 1
    if input1.stride(0) != output.stride(0) or input1.stride(1) != output.stride(1):
        return 1
    b, _, h, w = input1.shape
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_flowprojection_forward(_ptr(input1), _ptr(count), _ptr(output), b, h, w,
                                                        int(fillhole), _st(input1), _st(count), _stream(input1)))


def flowprojection_forward_batch(inputs1, counts, outputs, fillhole, inputs2=None):
    """The list form of flowprojection_forward / depthflowprojection_forward (inputs2: one depth tensor per item, or one
    tensor shared by all): every item in one launch triple.  Items share shape and strides."""
    n = len(inputs1)
    if n == 0 or len(counts) != n or len(outputs) != n:
        return 1
    if inputs2 is not None and not isinstance(inputs2, (list, tuple)):
        inputs2 = [inputs2] * n
    if inputs2 is not None and len(inputs2) != n:
        return 1
    f0, c0 = inputs1[0], counts[0]
    b, _, h, w = f0.shape
    for i in range(n):
        fl, cn, out = inputs1[i], counts[i], outputs[i]
        if fl.size(1) != 2 or tuple(fl.shape) != tuple(f0.shape) or tuple(out.shape) != tuple(f0.shape) or tuple(cn.shape) != (b, 1, h, w):
            return 1
        if not _same_strides(fl, f0) or not _same_strides(out, f0) or not _same_strides(cn, c0):
            return 1
        _dev(fl), _dev(cn), _dev(out)
        if inputs2 is not None:
            d = inputs2[i]
            if tuple(d.shape) != (b, 1, h, w) or not _same_strides(d, inputs2[0]):
                return 1
            _dev(d)
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])     # noqa: E731
    with torch.cuda.device(_dev(f0)):
        if inputs2 is None:
            return _finish(lib().vfi_flowprojection_forward_batch(arr(inputs1), arr(counts), arr(outputs), n, b, h, w, int(fillhole),
                                                                  _st(f0), _st(c0), _stream(f0)))
        return _finish(lib().vfi_depthflowprojection_forward_batch(arr(inputs1), arr(inputs2), arr(counts), arr(outputs), n, b, h, w,
                                                                   int(fillhole), _st(f0), _st(inputs2[0]), _st(c0), _stream(f0)))


def projection_reserve(batch, h, w, device=None):
    """Size the projection workspace of the current stream for [batch, *, h, w] frames (call before graph capture)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        return _finish(lib().vfi_projection_reserve(int(batch), int(h), int(w),
                                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


def release_workspaces():
    """Free every workspace the library holds (synchronises).  Only when no launch or graph still needs them."""
    return _finish(lib().vfi_release_workspaces())


def flowprojection_backward(input1, count, gradoutput, gradinput1):
    b, _, h, w = input1.shape
    if input1.size(1) != 2 or tuple(count.shape) != (b, 1, h, w):
        return 1
    if input1.stride(0) != gradinput1.stride(0) or input1.stride(1) != gradinput1.stride(1):
        return 1
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_flowprojection_backward(_ptr(input1), _ptr(count), _ptr(gradoutput),
                                                         _ptr(gradinput1), b, h, w, _st(input1), _st(count),
                                                         _stream(input1)))


def depthflowprojection_forward(input1, input2, count, output, fillhole):
    if input1.size(1) != 2 or input2.size(1) != 1:
        return 1
    if input1.stride(0) != output.stride(0) or input1.stride(1) != output.stride(1):
        return 1
    b, _, h, w = input1.shape
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_depthflowprojection_forward(
            _ptr(input1), _ptr(input2), _ptr(count), _ptr(output), b, h, w, int(fillhole), _st(input1), _st(input2),
            _st(count), _stream(input1)))


def depthflowprojection_backward(input1, input2, count, output, gradoutput, gradinput1, gradinput2):
    b, _, h, w = input1.shape
    if input1.size(1) != 2 or input2.size(1) != 1 or tuple(count.shape) != (b, 1, h, w):
        return 1
    if input1.stride(0) != gradinput1.stride(0) or input1.stride(1) != gradinput1.stride(1):
        return 1
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_depthflowprojection_backward(
            _ptr(input1), _ptr(input2), _ptr(count), _ptr(output), _ptr(gradoutput), _ptr(gradinput1),
            _ptr(gradinput2), b, h, w, _st(input1), _st(input2), _st(count), _stream(input1)))


def mindepthflowprojection_forward(input1, input2, count, output, fillhole):
    """mindepthflowprojection_cuda.cc:12-66; count and output zero-filled by the caller."""
    if input1.size(1) != 2 or input2.size(1) != 1:
        return 1
    if input1.stride(0) != output.stride(0) or input1.stride(1) != output.stride(1):
        return 1
    b, _, h, w = input1.shape
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_mindepthflowprojection_forward(
            _ptr(input1), _ptr(input2), _ptr(count), _ptr(output), b, h, w, int(fillhole), _st(input1), _st(input2),
            _st(count), _stream(input1)))


def mindepthflowprojection_backward(input1, input2, count, output, gradoutput, gradinput1, gradinput2):
    """mindepthflowprojection_cuda.cc:68-139; `output` and `gradinput2` are accepted and unused, as in the reference."""
    b, _, h, w = input1.shape
    if input1.size(1) != 2 or input2.size(1) != 1 or tuple(count.shape) != (b, 1, h, w):
        return 1
    if input1.stride(0) != gradinput1.stride(0) or input1.stride(1) != gradinput1.stride(1):
        return 1
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_mindepthflowprojection_backward(
            _ptr(input1), _ptr(input2), _ptr(count), _ptr(gradoutput), _ptr(gradinput1), b, h, w, _st(input1),
            _st(input2), _st(count), _stream(input1)))


# ---------------------------------------------------------------- interpolation_cuda / interpolationch_cuda

def interpolation_forward(input1, input2, output, require_c3=False):
    b, c, h, w = input1.shape
    if require_c3 and c != 3:
        return 1
    if tuple(input2.shape) != (b, 2, h, w):
        return 1
    if input1.stride(0) != output.stride(0) or input1.stride(1) != output.stride(1):
        return 1
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_interpolation_forward(_ptr(input1), _ptr(input2), _ptr(output), b, c, h, w,
                                                       _st(input1), _st(input2), _stream(input1)))


def interpolation_backward(input1, input2, gradoutput, gradinput1, gradinput2, require_c3=False):
    b, c, h, w = input1.shape
    if require_c3 and c != 3:
        return 1
    if tuple(input2.shape) != (b, 2, h, w):
        return 1
    if input1.stride(0) != gradinput1.stride(0) or input1.stride(1) != gradinput1.stride(1):
        return 1
    if input2.stride(0) != gradinput2.stride(0) or input2.stride(1) != gradinput2.stride(1):
        return 1
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_interpolation_backward(_ptr(input1), _ptr(input2), _ptr(gradoutput),
                                                        _ptr(gradinput1), _ptr(gradinput2), b, c, h, w, _st(input1),
                                                        _st(input2), _stream(input1)))


# ---------------------------------------------------------------- separableconv_cuda / separableconvflow_cuda

def _sep_checks(input1, input2, input3):
    b, c, h, w = input1.shape
    fs = input2.size(1)
    if c != 3 or input2.size(0) != b or input2.size(1) != input3.size(1):
        return None
    if input2.size(2) != h - fs + 1 or input2.size(3) != w - fs + 1:
        return None
    if input1.stride(3) != 1 or input2.stride(3) != 1 or input3.stride(3) != 1:
        return None
    if input2.stride(0) != input3.stride(0) or input2.stride(1) != input3.stride(1):
        return None
    return b, c, h, w, fs


def separableconv_forward(input1, input2, input3, output):
    dims = _sep_checks(input1, input2, input3)
    if dims is None or output.stride(3) != 1:
        return 1
    b, c, h, w, fs = dims
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_separableconv_forward(_ptr(input1), _ptr(input2), _ptr(input3), _ptr(output), b, c,
                                                       h, w, fs, _st(input1), _st(input2), _st(input3), _st(output),
                                                       _stream(input1)))


def separableconv_backward(input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3):
    dims = _sep_checks(input1, input2, input3)
    if dims is None or gradoutput.stride(3) != 1:
        return 1
    b, c, h, w, fs = dims
    with torch.cuda.device(_dev(input1)):
        return _finish(lib().vfi_separableconv_backward(
            _ptr(input1), _ptr(input2), _ptr(input3), _ptr(gradoutput), _ptr(gradinput1), _ptr(gradinput2),
            _ptr(gradinput3), b, c, h, w, fs, _st(input1), _st(input2), _st(input3), _st(gradoutput),
            _stream(input1)))


def separableconvflow_forward(input1, input2, input3, flow_output):
    dims = _sep_checks(input1, input2, input3)
    if dims is None or flow_output.stride(3) != 1:
        return 1
    b, c, h, w, fs = dims
    with torch.cuda.device(_dev(input2)):
        return _finish(lib().vfi_separableconvflow_forward(_ptr(input2), _ptr(input3), _ptr(flow_output), b, h, w,
                                                           fs, _st(input2), _st(input3), _st(flow_output),
                                                           _stream(input2)))


def separableconvflow_backward(input1, input2, input3, gradflow_output, gradinput2, gradinput3):
    dims = _sep_checks(input1, input2, input3)
    if dims is None or gradflow_output.stride(3) != 1:
        return 1
    b, c, h, w, fs = dims
    with torch.cuda.device(_dev(input2)):
        return _finish(lib().vfi_separableconvflow_backward(
            _ptr(input2), _ptr(input3), _ptr(gradflow_output), _ptr(gradinput2), _ptr(gradinput3), b, h, w, fs,
            _st(input2), _st(input3), _st(gradflow_output), _stream(input2)))


# ---------------------------------------------------------------- correlation_cuda

def correlation_output_dims(h, w, pad_size, kernel_size, max_displacement, stride1, stride2):
    oc, oh, ow = _i(), _i(), _i()
    err = lib().vfi_correlation_output_dims(h, w, pad_size, kernel_size, max_displacement, stride1, stride2,
                                            ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow))
    if err != 0:
        raise RuntimeError("CUDA call failed")
    return oc.value, oh.value, ow.value


def correlation_forward(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """Allocates and returns the output, as the reference binding resizes its `output` argument."""
    input1, input2 = input1.contiguous(), input2.contiguous()
    b, c, h, w = input1.shape
    oc, oh, ow = correlation_output_dims(h, w, pad_size, kernel_size, max_displacement, stride1, stride2)
    if input1.dtype == torch.float16:                       # the reference's at::Half instantiation
        _dev(input1, torch.float16), _dev(input2, torch.float16)
        output = torch.empty((b, oc, oh, ow), dtype=torch.float16, device=input1.device)
        with torch.cuda.device(input1.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(input1.device).cuda_stream)
            err = lib().vfi_correlation_forward_f16(_ptr(input1), _ptr(input2), _ptr(output), b, c, h, w, pad_size,
                                                    kernel_size, max_displacement, stride1, stride2, stream)
        if err != 0:
            raise RuntimeError("CUDA call failed")
        return output
    output = torch.empty((b, oc, oh, ow), dtype=torch.float32, device=input1.device)
    with torch.cuda.device(_dev(input1)):
        err = lib().vfi_correlation_forward(_ptr(input1), _ptr(input2), _ptr(output), b, c, h, w, pad_size,
                                            kernel_size, max_displacement, stride1, stride2, _stream(input1))
    if err != 0:
        raise RuntimeError("CUDA call failed")
    return output


def correlation_forward_pair(a1, a2, b1, b2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """(correlation_forward(a1, a2, ...), correlation_forward(b1, b2, ...)) from one launch: float32, equal shapes."""
    for t in (a1, a2, b1, b2):
        _dev(t)
        if tuple(t.shape) != tuple(a1.shape):
            raise RuntimeError("correlation_forward_pair: the four inputs must have one shape")
    a1, a2, b1, b2 = (t.contiguous() for t in (a1, a2, b1, b2))
    b, c, h, w = a1.shape
    oc, oh, ow = correlation_output_dims(h, w, pad_size, kernel_size, max_displacement, stride1, stride2)
    outa = torch.empty((b, oc, oh, ow), device=a1.device, dtype=torch.float32)
    outb = torch.empty_like(outa)
    with torch.cuda.device(_dev(a1)):
        err = lib().vfi_correlation_forward_pair(_ptr(a1), _ptr(a2), _ptr(outa), _ptr(b1), _ptr(b2), _ptr(outb), b, c, h, w,
                                                 pad_size, kernel_size, max_displacement, stride1, stride2, _stream(a1))
    if _finish(err) != 0:
        raise RuntimeError("correlation_forward_pair: the binding returned %d" % err)
    return outa, outb


def correlation_backward(input1, input2, gradoutput, pad_size, kernel_size, max_displacement, stride1, stride2):
    input1, input2, gradoutput = input1.contiguous(), input2.contiguous(), gradoutput.contiguous()
    b, c, h, w = input1.shape
    g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
    with torch.cuda.device(_dev(input1)):
        err = lib().vfi_correlation_backward(_ptr(input1), _ptr(input2), _ptr(gradoutput), _ptr(g1), _ptr(g2), b, c,
                                             h, w, pad_size, kernel_size, max_displacement, stride1, stride2,
                                             _stream(input1))
    if err != 0:
        raise RuntimeError("CUDA call failed")
    return g1, g2


# ---------------------------------------------------------------- glue either side of the ops

def _nchw_ok(*tensors):
    return all(t.dim() == 4 and t.stride(3) == 1 for t in tensors)


def flow_upsample4(input, output, mul0, mul1):
    b, c, hq, wq = input.shape
    if not _nchw_ok(input, output) or tuple(output.shape) != (b, c, 4 * hq, 4 * wq):
        return 1
    _dev(output)
    with torch.cuda.device(_dev(input)):
        return _finish(lib().vfi_flow_upsample4(_ptr(input), _ptr(output), b, c, hq, wq, mul0, mul1, _st(input),
                                                _st(output), _stream(input)))


def flowprojection_forward_up4(flow_q, count, output, mul0, mul1, fillhole):
    b, c, hq, wq = flow_q.shape
    if c != 2 or not _nchw_ok(flow_q, count, output):
        return 1
    if tuple(output.shape) != (b, 2, 4 * hq, 4 * wq) or tuple(count.shape) != (b, 1, 4 * hq, 4 * wq):
        return 1
    _dev(count), _dev(output)
    with torch.cuda.device(_dev(flow_q)):
        return _finish(lib().vfi_flowprojection_forward_up4(_ptr(flow_q), _ptr(count), _ptr(output), b, hq, wq, mul0,
                                                            mul1, int(fillhole), _st(flow_q), _st(count), _st(output),
                                                            _stream(flow_q)))


def depthflowprojection_forward_up4(flow_q, input2, count, output, mul0, mul1, fillhole):
    b, c, hq, wq = flow_q.shape
    if c != 2 or not _nchw_ok(flow_q, input2, count, output):
        return 1
    full = (b, 1, 4 * hq, 4 * wq)
    if tuple(output.shape) != (b, 2, 4 * hq, 4 * wq) or tuple(count.shape) != full or tuple(input2.shape) != full:
        return 1
    _dev(input2), _dev(count), _dev(output)
    with torch.cuda.device(_dev(flow_q)):
        return _finish(lib().vfi_depthflowprojection_forward_up4(
            _ptr(flow_q), _ptr(input2), _ptr(count), _ptr(output), b, hq, wq, mul0, mul1, int(fillhole), _st(flow_q),
            _st(input2), _st(count), _st(output), _stream(flow_q)))


def _same_strides(a, b):
    """same layout: the stride of a dimension of size 1 is never used (torch leaves it arbitrary, e.g. after slicing)"""
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n != 1)


def filterinterp_blend_forward(ref0, ref2, flow0, flow2, filt0, filt2, blend, out0, out2, w0, w2):
    """out0 / out2 may be None."""
    dims = _fi_checks(ref0, flow0, filt0, None)
    if dims is None or not (_same_strides(ref0, ref2) and _same_strides(flow0, flow2) and _same_strides(filt0, filt2)):
        return 1
    outs = [t for t in (out0, out2) if t is not None]
    if blend.shape != ref0.shape or blend.stride(3) != 1 or not all(_same_strides(blend, t) for t in outs):
        return 1
    b, c, h, w = dims
    for t in (ref2, flow0, flow2, filt0, filt2, blend, *outs):
        _dev(t)
    null = ctypes.c_void_p(0)
    with torch.cuda.device(_dev(ref0)):
        return _finish(lib().vfi_filterinterp_blend_forward(
            _ptr(ref0), _ptr(ref2), _ptr(flow0), _ptr(flow2), _ptr(filt0), _ptr(filt2), _ptr(blend),
            _ptr(out0) if out0 is not None else null, _ptr(out2) if out2 is not None else null, b, c, h, w,
            filt0.size(1), w0, w2, _st(ref0), _st(flow0), _st(filt0), _st(blend), _stream(ref0)))


def pwc_warp_forward(x, flow, output, align_corners=True):
    b, c, h, w = x.shape
    if not _nchw_ok(x, flow, output) or tuple(flow.shape) != (b, 2, h, w) or output.shape != x.shape:
        return 1
    _dev(flow), _dev(output)
    with torch.cuda.device(_dev(x)):
        return _finish(lib().vfi_pwc_warp_forward(_ptr(x), _ptr(flow), _ptr(output), b, c, h, w, int(bool(align_corners)),
                                                  _st(x), _st(flow), _st(output), _stream(x)))


def pwc_warp_correlation_forward(input1, input2, flow, align_corners=True):
    """correlation(input1, warp(input2, flow)) of PWC-Net (pad 4, k 1, md 4, strides 1) in one launch; returns [B,81,h,w]."""
    input1, input2 = input1.contiguous(), input2.contiguous()
    b, c, h, w = input1.shape
    if tuple(input2.shape) != (b, c, h, w) or tuple(flow.shape) != (b, 2, h, w) or flow.stride(3) != 1:
        raise RuntimeError("pwc_warp_correlation_forward: shape mismatch")
    _dev(input2), _dev(flow)
    output = torch.empty((b, 81, h, w), dtype=torch.float32, device=input1.device)
    with torch.cuda.device(_dev(input1)):
        err = _finish(lib().vfi_pwc_warp_correlation_forward(_ptr(input1), _ptr(input2), _ptr(flow), _ptr(output), b, c, h, w,
                                                             int(bool(align_corners)), _st(flow), _stream(input1)))
    if err != 0:
        raise RuntimeError("CUDA call failed")
    return output


def frame_u8_to_planar(src_hwc, dst, pad_left, pad_right, pad_top, pad_bottom):
    """src_hwc: dense uint8 [B,h,w,3]; dst: float32 [B,3,h+pt+pb,w+pl+pr]."""
    b, h, w, c = src_hwc.shape
    if c != 3 or not src_hwc.is_contiguous() or dst.stride(3) != 1:
        return 1
    if tuple(dst.shape) != (b, 3, h + pad_top + pad_bottom, w + pad_left + pad_right):
        return 1
    _dev(dst)
    with torch.cuda.device(_dev(src_hwc, torch.uint8)):
        return _finish(lib().vfi_frame_u8_to_planar(_ptr(src_hwc), _ptr(dst), b, h, w, pad_left, pad_right, pad_top,
                                                    pad_bottom, _st(dst), _stream(dst)))


def planar_to_frame_u8(src, dst_hwc, top, left):
    """src: float32 [B,3,H,W]; dst_hwc: dense uint8 [B,h,w,3], the crop at (top, left)."""
    b, h, w, c = dst_hwc.shape
    if c != 3 or not dst_hwc.is_contiguous() or src.stride(3) != 1 or src.size(0) != b or src.size(1) != 3:
        return 1
    if top < 0 or left < 0 or top + h > src.size(2) or left + w > src.size(3):
        return 1
    _dev(dst_hwc, torch.uint8)
    with torch.cuda.device(_dev(src)):
        return _finish(lib().vfi_planar_to_frame_u8(_ptr(src), _ptr(dst_hwc), b, h, w, top, left, _st(src),
                                                    _stream(src)))


def frame_error_sums(a, b, sums):
    """a, b: dense uint8 tensors of one shape; sums: int64[2] on the GPU, zeroed by the caller."""
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous() or sums.numel() < 2:
        return 1
    _dev(b, torch.uint8), _dev(sums, torch.int64)
    with torch.cuda.device(_dev(a, torch.uint8)):
        stream = ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        return _finish(lib().vfi_frame_error_sums(_ptr(a), _ptr(b), a.numel(), _ptr(sums), stream))


def frame_ssim_sums(a, b, sums):
    """a, b: dense uint8 [B,h,w,3]; sums: int64[1] on the GPU, zeroed by the caller (2^-32 fixed point)."""
    if a.shape != b.shape or a.dim() != 4 or a.size(3) != 3 or not a.is_contiguous() or not b.is_contiguous():
        return 1
    if sums.numel() < 1:
        return 1
    _dev(b, torch.uint8), _dev(sums, torch.int64)
    with torch.cuda.device(_dev(a, torch.uint8)):
        stream = ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)
        return _finish(lib().vfi_frame_ssim_sums(_ptr(a), _ptr(b), a.size(0), a.size(1), a.size(2), _ptr(sums), stream))
