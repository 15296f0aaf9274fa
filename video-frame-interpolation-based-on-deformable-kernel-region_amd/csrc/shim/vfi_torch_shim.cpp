// vfi_torch_shim.cpp -- the reference's eight pybind11 extension modules, re-exported
// on top of the C ABI of libvfi_hip.so (include/vfi_hip.h).
//
// Module and function names, positional signatures, return values and error
// behaviour are those of the reference bindings (file:line cited per function),
// so `import filterinterpolation_cuda as my_lib` etc. in DAIN-style wrappers work
// unchanged.  This file contains no kernels and no arithmetic: it checks shapes
// and strides exactly as the reference .cc files do (silent `return 1` on a
// mismatch), picks the current HIP stream of the tensors' device, and forwards
// raw pointers.  One translation unit defines all eight PyInit_* symbols; the
// build copies the resulting shared object under the eight module names.
#include <torch/extension.h>

// torch-ROCm tensors carry DeviceType::CUDA ("masquerading"): the guard and the stream
// accessor have to be the *MasqueradingAsCUDA flavours, the plain c10::hip ones reject them
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cmath>

#include "../../../include/vfi_hip.h"

namespace {

inline vfi_strides st(const at::Tensor& t) { return vfi_strides{t.stride(0), t.stride(1), t.stride(2)}; }
inline const float* cptr(const at::Tensor& t) { return t.data_ptr<float>(); }
inline float* mptr(at::Tensor& t) { return t.data_ptr<float>(); }

struct StreamScope {
    c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard;
    vfi_stream_t stream;
    explicit StreamScope(const at::Tensor& t, bool allow_half = false) {
        TORCH_CHECK(t.is_cuda(), "expected a GPU tensor");
        TORCH_CHECK(t.scalar_type() == at::kFloat || (allow_half && t.scalar_type() == at::kHalf), "expected float32 tensors");
        guard.set_device(t.device());
        stream = (vfi_stream_t)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
    }
};

// launch failure -> the reference's AT_ERROR("CUDA call failed"); shape error -> return 1
inline int finish(int err) {
    TORCH_CHECK(err != VFI_ERR_LAUNCH, "CUDA call failed");
    return err;
}

// ---------------------------------------------------------------- filterinterpolation_cuda

// filterinterpolation_cuda.cc:537-606
int FilterInterpolationLayer_gpu_forward_ori(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                             at::Tensor& output) {
    const int error = 1;
    const int channel = input1.size(1), batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != 2) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h) return error;
    if (input2.size(3) != w) return error;
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1) return error;
    if (input1.stride(0) != output.stride(0)) return error;
    if (input1.stride(1) != output.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_filterinterp_forward_ori(cptr(input1), cptr(input2), cptr(input3), mptr(output), batch, channel,
                                               h, w, (int)input3.size(1), st(input1), st(input2), st(input3),
                                               s.stream));
}

// filterinterpolation_cuda.cc:608-687
int FilterInterpolationLayer_gpu_backward_ori(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                              at::Tensor& gradoutput, at::Tensor& gradinput1,
                                              at::Tensor& gradinput2, at::Tensor& gradinput3) {
    const int error = 1;
    const int channel = input1.size(1), batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != 2) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h) return error;
    if (input2.size(3) != w) return error;
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1) return error;
    if (input1.stride(0) != gradinput1.stride(0)) return error;
    if (input2.stride(0) != gradinput2.stride(0)) return error;
    if (input1.stride(1) != gradinput1.stride(1)) return error;
    if (input2.stride(1) != gradinput2.stride(1)) return error;
    if (input3.stride(1) != gradinput3.stride(1)) return error;
    // the kernels address gradoutput with input1's strides and gradinput3 with input3's, as the reference does
    StreamScope s(input1);
    return finish(vfi_filterinterp_backward_ori(cptr(input1), cptr(input2), cptr(input3), cptr(gradoutput),
                                                mptr(gradinput1), mptr(gradinput2), mptr(gradinput3), batch, channel,
                                                h, w, (int)input3.size(1), st(input1), st(input2), st(input3),
                                                s.stream));
}

// filterinterpolation_cuda.cc:11-92 (variant 0) and :191-272 (variant 1)
int defor4(int variant, at::Tensor& input1, at::Tensor& input2, at::Tensor& input3, at::Tensor& input4,
           at::Tensor& output) {
    const int error = 1;
    const int channel = input1.size(1), batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != 2) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h) return error;
    if (input2.size(3) != w) return error;
    const int filter_size = (int)std::sqrt((float)input3.size(1));
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1 || input4.stride(3) != 1)
        return error;
    StreamScope s(input1);
    return finish(vfi_filterinterp_forward_defor(variant, cptr(input1), cptr(input2), cptr(input3), cptr(input4),
                                                 mptr(output), batch, channel, h, w, filter_size, st(input1),
                                                 st(input2), st(input3), st(input4), s.stream));
}
int FilterInterpolationLayer_gpu_forward(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                         at::Tensor& input4, at::Tensor& output) {
    return defor4(VFI_DEFOR_OFFSET, input1, input2, input3, input4, output);
}
int FilterInterpolationLayer_gpu_forward_deforconv(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                                   at::Tensor& input4, at::Tensor& output) {
    return defor4(VFI_DEFOR_REGION, input1, input2, input3, input4, output);
}

// filterinterpolation_cuda.cc:374-447
int FilterInterpolationLayer_gpu_forward_nofilterwithdeforconv(at::Tensor& input1, at::Tensor& input2,
                                                               at::Tensor& input3, at::Tensor& output) {
    const int error = 1;
    const int channel = input1.size(1), batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != 2) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h) return error;
    if (input2.size(3) != w) return error;
    const int filter_size = (int)std::sqrt((float)(input3.size(1) / 2));
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1) return error;
    if (input1.stride(0) != output.stride(0)) return error;
    if (input1.stride(1) != output.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_filterinterp_forward_defor(VFI_DEFOR_NOFILTER, cptr(input1), cptr(input2), cptr(input3),
                                                 nullptr, mptr(output), batch, channel, h, w, filter_size, st(input1),
                                                 st(input2), st(input3), st(input3), s.stream));
}

// filterinterpolation_cuda.cc:93-187 (variant 0) and :273-367 (variant 1)
int defor4_bwd(int variant, at::Tensor& input1, at::Tensor& input2, at::Tensor& input3, at::Tensor& input4,
               at::Tensor& gradoutput, at::Tensor& gradinput1, at::Tensor& gradinput2, at::Tensor& gradinput3,
               at::Tensor& gradinput4) {
    const int error = 1;
    const int channel = input1.size(1), batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != 2) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h) return error;
    if (input2.size(3) != w) return error;
    const int filter_size = (int)std::sqrt((float)input3.size(1));
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1 || input4.stride(3) != 1)
        return error;
    if (input1.stride(0) != gradinput1.stride(0)) return error;
    if (input2.stride(0) != gradinput2.stride(0)) return error;
    if (input1.stride(1) != gradinput1.stride(1)) return error;
    if (input2.stride(1) != gradinput2.stride(1)) return error;
    if (input3.stride(1) != gradinput3.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_filterinterp_backward_defor(variant, cptr(input1), cptr(input2), cptr(input3), cptr(input4),
                                                  cptr(gradoutput), mptr(gradinput1), mptr(gradinput2),
                                                  mptr(gradinput3), mptr(gradinput4), batch, channel, h, w,
                                                  filter_size, st(input1), st(input2), st(input3), st(input4),
                                                  s.stream));
}
int FilterInterpolationLayer_gpu_backward(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                          at::Tensor& input4, at::Tensor& gradoutput, at::Tensor& gradinput1,
                                          at::Tensor& gradinput2, at::Tensor& gradinput3, at::Tensor& gradinput4) {
    return defor4_bwd(VFI_DEFOR_OFFSET, input1, input2, input3, input4, gradoutput, gradinput1, gradinput2,
                      gradinput3, gradinput4);
}
int FilterInterpolationLayer_gpu_backward_deforconv(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                                    at::Tensor& input4, at::Tensor& gradoutput,
                                                    at::Tensor& gradinput1, at::Tensor& gradinput2,
                                                    at::Tensor& gradinput3, at::Tensor& gradinput4) {
    return defor4_bwd(VFI_DEFOR_REGION, input1, input2, input3, input4, gradoutput, gradinput1, gradinput2,
                      gradinput3, gradinput4);
}

// filterinterpolation_cuda.cc:448-533
int FilterInterpolationLayer_gpu_backward_nofilterwithdeforconv(at::Tensor& input1, at::Tensor& input2,
                                                                at::Tensor& input3, at::Tensor& gradoutput,
                                                                at::Tensor& gradinput1, at::Tensor& gradinput2,
                                                                at::Tensor& gradinput3) {
    const int error = 1;
    const int channel = input1.size(1), batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != 2) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h) return error;
    if (input2.size(3) != w) return error;
    const int filter_size = (int)std::sqrt((float)(input3.size(1) / 2));
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1) return error;
    if (input1.stride(0) != gradinput1.stride(0)) return error;
    if (input2.stride(0) != gradinput2.stride(0)) return error;
    if (input1.stride(1) != gradinput1.stride(1)) return error;
    if (input2.stride(1) != gradinput2.stride(1)) return error;
    if (input3.stride(1) != gradinput3.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_filterinterp_backward_defor(VFI_DEFOR_NOFILTER, cptr(input1), cptr(input2), cptr(input3),
                                                  nullptr, cptr(gradoutput), mptr(gradinput1), mptr(gradinput2),
                                                  mptr(gradinput3), nullptr, batch, channel, h, w, filter_size,
                                                  st(input1), st(input2), st(input3), st(input3), s.stream));
}

// ---------------------------------------------------------------- flowprojection_cuda

// flowprojection_cuda.cc:9-57
int FlowProjectionLayer_gpu_forward(at::Tensor& input1, at::Tensor& count, at::Tensor& output, int fillhole) {
    const int error = 1;
    if (input1.size(1) != 2) return error;
    const int batch = input1.size(0), h = input1.size(2), w = input1.size(3);
    if (input1.stride(0) != output.stride(0)) return error;
    if (input1.stride(1) != output.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_flowprojection_forward(cptr(input1), mptr(count), mptr(output), batch, h, w, fillhole,
                                             st(input1), st(count), s.stream));
}

// flowprojection_cuda.cc:59-114
int FlowProjectionLayer_gpu_backward(at::Tensor& input1, at::Tensor& count, at::Tensor& gradoutput,
                                     at::Tensor& gradinput1) {
    const int error = 1;
    if (input1.size(1) != 2) return error;
    const int batch = input1.size(0);
    if (count.size(0) != batch) return error;
    if (count.size(1) != 1) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (count.size(2) != h) return error;
    if (count.size(3) != w) return error;
    if (input1.stride(0) != gradinput1.stride(0)) return error;
    if (input1.stride(1) != gradinput1.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_flowprojection_backward(cptr(input1), cptr(count), cptr(gradoutput), mptr(gradinput1), batch, h,
                                              w, st(input1), st(count), s.stream));
}

// ---------------------------------------------------------------- depthflowprojection_cuda

// depthflowprojection_cuda.cc:10-68
int DepthFlowProjectionLayer_gpu_forward(at::Tensor& input1, at::Tensor& input2, at::Tensor& count,
                                         at::Tensor& output, int fillhole) {
    const int error = 1;
    if (input1.size(1) != 2) return error;
    const int batch = input1.size(0), h = input1.size(2), w = input1.size(3);
    if (input2.size(1) != 1) return error;
    if (input1.stride(0) != output.stride(0)) return error;
    if (input1.stride(1) != output.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_depthflowprojection_forward(cptr(input1), cptr(input2), mptr(count), mptr(output), batch, h, w,
                                                  fillhole, st(input1), st(input2), st(count), s.stream));
}

// depthflowprojection_cuda.cc:70-139
int DepthFlowProjectionLayer_gpu_backward(at::Tensor& input1, at::Tensor& input2, at::Tensor& count,
                                          at::Tensor& output, at::Tensor& gradoutput, at::Tensor& gradinput1,
                                          at::Tensor& gradinput2) {
    const int error = 1;
    if (input1.size(1) != 2) return error;
    const int batch = input1.size(0);
    if (count.size(0) != batch) return error;
    if (count.size(1) != 1) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(1) != 1) return error;
    if (count.size(2) != h) return error;
    if (count.size(3) != w) return error;
    if (input1.stride(0) != gradinput1.stride(0)) return error;
    if (input1.stride(1) != gradinput1.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_depthflowprojection_backward(cptr(input1), cptr(input2), cptr(count), cptr(output),
                                                   cptr(gradoutput), mptr(gradinput1), mptr(gradinput2), batch, h, w,
                                                   st(input1), st(input2), st(count), s.stream));
}

// ---------------------------------------------------------------- mindepthflowprojection_cuda

// mindepthflowprojection_cuda.cc:12-66
int minDepthFlowProjectionLayer_gpu_forward(at::Tensor& input1, at::Tensor& input2, at::Tensor& count,
                                            at::Tensor& output, int fillhole) {
    const int error = 1;
    if (input1.size(1) != 2) return error;
    const int batch = input1.size(0), h = input1.size(2), w = input1.size(3);
    if (input2.size(1) != 1) return error;
    if (input1.stride(0) != output.stride(0)) return error;
    if (input1.stride(1) != output.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_mindepthflowprojection_forward(cptr(input1), cptr(input2), mptr(count), mptr(output), batch, h, w,
                                                     fillhole, st(input1), st(input2), st(count), s.stream));
}

// mindepthflowprojection_cuda.cc:68-139 (output and gradinput2 are passed and never used by the reference kernel)
int minDepthFlowProjectionLayer_gpu_backward(at::Tensor& input1, at::Tensor& input2, at::Tensor& count,
                                             at::Tensor& output, at::Tensor& gradoutput, at::Tensor& gradinput1,
                                             at::Tensor& gradinput2) {
    const int error = 1;
    if (input1.size(1) != 2) return error;
    const int batch = input1.size(0);
    if (count.size(0) != batch) return error;
    if (count.size(1) != 1) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(1) != 1) return error;
    if (count.size(2) != h) return error;
    if (count.size(3) != w) return error;
    if (input1.stride(0) != gradinput1.stride(0)) return error;
    if (input1.stride(1) != gradinput1.stride(1)) return error;
    (void)output; (void)gradinput2;
    StreamScope s(input1);
    return finish(vfi_mindepthflowprojection_backward(cptr(input1), cptr(input2), cptr(count), cptr(gradoutput),
                                                      mptr(gradinput1), batch, h, w, st(input1), st(input2), st(count),
                                                      s.stream));
}

// ---------------------------------------------------------------- interpolation_cuda / interpolationch_cuda

// interpolation_cuda.cc:10-60 ; interpolationch_cuda.cc drops the channel==3 test (:19)
int interp_fwd(bool require_c3, at::Tensor& input1, at::Tensor& input2, at::Tensor& output) {
    const int error = 1;
    const int channel = input1.size(1);
    if (require_c3 && channel != 3) return error;
    const int batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != 2) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h) return error;
    if (input2.size(3) != w) return error;
    if (input1.stride(0) != output.stride(0)) return error;
    if (input1.stride(1) != output.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_interpolation_forward(cptr(input1), cptr(input2), mptr(output), batch, channel, h, w,
                                            st(input1), st(input2), s.stream));
}
// interpolation_cuda.cc:63-121
int interp_bwd(bool require_c3, at::Tensor& input1, at::Tensor& input2, at::Tensor& gradoutput,
               at::Tensor& gradinput1, at::Tensor& gradinput2) {
    const int error = 1;
    const int channel = input1.size(1);
    if (require_c3 && channel != 3) return error;
    const int batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != 2) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h) return error;
    if (input2.size(3) != w) return error;
    if (input1.stride(0) != gradinput1.stride(0)) return error;
    if (input2.stride(0) != gradinput2.stride(0)) return error;
    if (input1.stride(1) != gradinput1.stride(1)) return error;
    if (input2.stride(1) != gradinput2.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_interpolation_backward(cptr(input1), cptr(input2), cptr(gradoutput), mptr(gradinput1),
                                             mptr(gradinput2), batch, channel, h, w, st(input1), st(input2),
                                             s.stream));
}
int InterpolationLayer_gpu_forward(at::Tensor& a, at::Tensor& b, at::Tensor& o) { return interp_fwd(true, a, b, o); }
int InterpolationLayer_gpu_backward(at::Tensor& a, at::Tensor& b, at::Tensor& g, at::Tensor& g1, at::Tensor& g2) {
    return interp_bwd(true, a, b, g, g1, g2);
}
int InterpolationChLayer_gpu_forward(at::Tensor& a, at::Tensor& b, at::Tensor& o) { return interp_fwd(false, a, b, o); }
int InterpolationChLayer_gpu_backward(at::Tensor& a, at::Tensor& b, at::Tensor& g, at::Tensor& g1, at::Tensor& g2) {
    return interp_bwd(false, a, b, g, g1, g2);
}

// ---------------------------------------------------------------- separableconv_cuda

// separableconv_cuda.cc:10-87
int SeparableConvLayer_gpu_forward(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3, at::Tensor& output) {
    const int error = 1;
    const int channel = input1.size(1);
    if (channel != 3) return error;
    const int batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != input3.size(1)) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h - input2.size(1) + 1) return error;
    if (input2.size(3) != w - input2.size(1) + 1) return error;
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1 || output.stride(3) != 1)
        return error;
    if (input2.stride(0) != input3.stride(0)) return error;
    if (input2.stride(1) != input3.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_separableconv_forward(cptr(input1), cptr(input2), cptr(input3), mptr(output), batch, channel, h,
                                            w, (int)input2.size(1), st(input1), st(input2), st(input3), st(output),
                                            s.stream));
}

// separableconv_cuda.cc:88-174
int SeparableConvLayer_gpu_backward(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                    at::Tensor& gradoutput, at::Tensor& gradinput1, at::Tensor& gradinput2,
                                    at::Tensor& gradinput3) {
    const int error = 1;
    const int channel = input1.size(1);
    if (channel != 3) return error;
    const int batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    if (input2.size(1) != input3.size(1)) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h - input2.size(1) + 1) return error;
    if (input2.size(3) != w - input2.size(1) + 1) return error;
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1 || gradoutput.stride(3) != 1)
        return error;
    if (input1.stride(0) != gradinput1.stride(0)) return error;
    if (input2.stride(0) != gradinput2.stride(0)) return error;
    if (input1.stride(1) != gradinput1.stride(1)) return error;
    if (input2.stride(1) != gradinput2.stride(1)) return error;
    if (input3.stride(1) != gradinput3.stride(1)) return error;
    StreamScope s(input1);
    return finish(vfi_separableconv_backward(cptr(input1), cptr(input2), cptr(input3), cptr(gradoutput),
                                             mptr(gradinput1), mptr(gradinput2), mptr(gradinput3), batch, channel, h,
                                             w, (int)input2.size(1), st(input1), st(input2), st(input3),
                                             st(gradoutput), s.stream));
}

// ---------------------------------------------------------------- separableconvflow_cuda

// separableconvflow_cuda.cc:9-102
int SeparableConvFlowLayer_gpu_forward(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                       at::Tensor& flow_output) {
    const int error = 1;
    if (input1.size(1) != 3) return error;
    const int batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h - input2.size(1) + 1) return error;
    if (input2.size(3) != w - input2.size(1) + 1) return error;
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1 || flow_output.stride(3) != 1)
        return error;
    if (input2.stride(0) != input3.stride(0)) return error;
    if (input2.stride(1) != input3.stride(1)) return error;
    StreamScope s(input2);
    return finish(vfi_separableconvflow_forward(cptr(input2), cptr(input3), mptr(flow_output), batch, h, w,
                                                (int)input2.size(1), st(input2), st(input3), st(flow_output),
                                                s.stream));
}

// separableconvflow_cuda.cc:103-199
int SeparableConvFlowLayer_gpu_backward(at::Tensor& input1, at::Tensor& input2, at::Tensor& input3,
                                        at::Tensor& gradflow_output, at::Tensor& gradinput1,
                                        at::Tensor& gradinput2, at::Tensor& gradinput3) {
    const int error = 1;
    if (input1.size(1) != 3) return error;
    const int batch = input1.size(0);
    if (input2.size(0) != batch) return error;
    const int h = input1.size(2), w = input1.size(3);
    if (input2.size(2) != h - input2.size(1) + 1) return error;
    if (input2.size(3) != w - input2.size(1) + 1) return error;
    if (input1.stride(3) != 1 || input2.stride(3) != 1 || input3.stride(3) != 1 || gradflow_output.stride(3) != 1)
        return error;
    if (input2.stride(0) != gradinput2.stride(0)) return error;
    if (input2.stride(1) != gradinput2.stride(1)) return error;
    if (input3.stride(1) != gradinput3.stride(1)) return error;
    (void)gradinput1;       // the reference never writes it: the image does not enter the flow
    StreamScope s(input2);
    return finish(vfi_separableconvflow_backward(cptr(input2), cptr(input3), cptr(gradflow_output), mptr(gradinput2),
                                                 mptr(gradinput3), batch, h, w, (int)input2.size(1), st(input2),
                                                 st(input3), st(gradflow_output), s.stream));
}

// ---------------------------------------------------------------- correlation_cuda

// correlation_cuda.cc:8-85.  The reference resizes and zero-fills the two padded
// NHWC repack buffers and the output.  The kernels here need no repack, so
// rInput1/rInput2 are resized to the reference's shapes (a caller inspecting
// them sees the same sizes) but not filled; every output element is written.
int correlation_forward(at::Tensor& input1, at::Tensor& input2, at::Tensor& rInput1, at::Tensor& rInput2,
                        at::Tensor& output, int pad_size, int kernel_size, int max_displacement, int stride1,
                        int stride2, int corr_type_multiply) {
    (void)corr_type_multiply;                   // accepted and ignored, as in the reference
    const int batch = input1.size(0), channel = input1.size(1), h = input1.size(2), w = input1.size(3);
    int oc = 0, oh = 0, ow = 0;
    TORCH_CHECK(vfi_correlation_output_dims(h, w, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh,
                                            &ow) == VFI_OK, "CUDA call failed");
    rInput1.resize_({batch, h + 2 * pad_size, w + 2 * pad_size, channel});
    rInput2.resize_({batch, h + 2 * pad_size, w + 2 * pad_size, channel});
    output.resize_({batch, oc, oh, ow});
    at::Tensor a = input1.contiguous(), b = input2.contiguous();   // the reference kernels assume dense NCHW
    StreamScope s(a, /*allow_half=*/true);
    int err;
    if (a.scalar_type() == at::kHalf) {         // AT_DISPATCH_FLOATING_TYPES_AND_HALF (correlation_cuda_kernel.cu:386, 403)
        TORCH_CHECK(b.scalar_type() == at::kHalf && output.scalar_type() == at::kHalf && a.is_cuda() && b.is_cuda() &&
                    output.is_cuda(), "correlation_cuda.forward: half inputs need a half output on the GPU");
        err = vfi_correlation_forward_f16(a.data_ptr(), b.data_ptr(), output.data_ptr(), batch, channel, h, w, pad_size,
                                          kernel_size, max_displacement, stride1, stride2, s.stream);
    } else {
        err = vfi_correlation_forward(cptr(a), cptr(b), mptr(output), batch, channel, h, w, pad_size,
                                      kernel_size, max_displacement, stride1, stride2, s.stream);
    }
    TORCH_CHECK(err == VFI_OK, "CUDA call failed");
    return 1;                                   // the reference binding always returns 1 (cc:83)
}

// correlation_cuda.cc:87-165
int correlation_backward(at::Tensor& input1, at::Tensor& input2, at::Tensor& rInput1, at::Tensor& rInput2,
                         at::Tensor& gradOutput, at::Tensor& gradInput1, at::Tensor& gradInput2, int pad_size,
                         int kernel_size, int max_displacement, int stride1, int stride2, int corr_type_multiply) {
    (void)corr_type_multiply;
    const int batch = input1.size(0), channel = input1.size(1), h = input1.size(2), w = input1.size(3);
    rInput1.resize_({batch, h + 2 * pad_size, w + 2 * pad_size, channel});
    rInput2.resize_({batch, h + 2 * pad_size, w + 2 * pad_size, channel});
    gradInput1.resize_({batch, channel, h, w});
    gradInput2.resize_({batch, channel, h, w});
    at::Tensor a = input1.contiguous(), b = input2.contiguous(), g = gradOutput.contiguous();
    StreamScope s(a);
    const int err = vfi_correlation_backward(cptr(a), cptr(b), cptr(g), mptr(gradInput1), mptr(gradInput2), batch,
                                             channel, h, w, pad_size, kernel_size, max_displacement, stride1, stride2,
                                             s.stream);
    TORCH_CHECK(err == VFI_OK, "CUDA call failed");
    return 1;
}

}  // namespace

PYBIND11_MODULE(filterinterpolation_cuda, m) {
    m.def("FilterInterpolationLayer_gpu_forward_ori", &FilterInterpolationLayer_gpu_forward_ori, "FilterInterpolation forward ori version (HIP, gfx950)");
    m.def("FilterInterpolationLayer_gpu_backward_ori", &FilterInterpolationLayer_gpu_backward_ori, "FilterInterpolation backward ori version (HIP, gfx950)");
    m.def("FilterInterpolationLayer_gpu_forward", &FilterInterpolationLayer_gpu_forward, "FilterInterpolation forward (HIP, gfx950)");
    m.def("FilterInterpolationLayer_gpu_forward_deforconv", &FilterInterpolationLayer_gpu_forward_deforconv, "FilterInterpolation forward deforconv (HIP, gfx950)");
    m.def("FilterInterpolationLayer_gpu_forward_nofilterwithdeforconv", &FilterInterpolationLayer_gpu_forward_nofilterwithdeforconv, "FilterInterpolation forward no filter with deforconv (HIP, gfx950)");
    m.def("FilterInterpolationLayer_gpu_backward", &FilterInterpolationLayer_gpu_backward, "FilterInterpolation backward (HIP, gfx950)");
    m.def("FilterInterpolationLayer_gpu_backward_deforconv", &FilterInterpolationLayer_gpu_backward_deforconv, "FilterInterpolation backward deforconv (HIP, gfx950)");
    m.def("FilterInterpolationLayer_gpu_backward_nofilterwithdeforconv", &FilterInterpolationLayer_gpu_backward_nofilterwithdeforconv, "FilterInterpolation backward no filter with deforconv (HIP, gfx950)");
}
PYBIND11_MODULE(flowprojection_cuda, m) {
    m.def("FlowProjectionLayer_gpu_forward", &FlowProjectionLayer_gpu_forward, "FlowProjection forward (HIP, gfx950)");
    m.def("FlowProjectionLayer_gpu_backward", &FlowProjectionLayer_gpu_backward, "FlowProjection backward (HIP, gfx950)");
}
PYBIND11_MODULE(depthflowprojection_cuda, m) {
    m.def("DepthFlowProjectionLayer_gpu_forward", &DepthFlowProjectionLayer_gpu_forward, "DepthFlowProjection forward (HIP, gfx950)");
    m.def("DepthFlowProjectionLayer_gpu_backward", &DepthFlowProjectionLayer_gpu_backward, "DepthFlowProjection backward (HIP, gfx950)");
}
PYBIND11_MODULE(mindepthflowprojection_cuda, m) {
    m.def("minDepthFlowProjectionLayer_gpu_forward", &minDepthFlowProjectionLayer_gpu_forward, "minDepthFlowProjection forward (HIP, gfx950)");
    m.def("minDepthFlowProjectionLayer_gpu_backward", &minDepthFlowProjectionLayer_gpu_backward, "minDepthFlowProjection backward (HIP, gfx950)");
}
PYBIND11_MODULE(interpolation_cuda, m) {
    m.def("InterpolationLayer_gpu_forward", &InterpolationLayer_gpu_forward, "Interpolation forward (HIP, gfx950)");
    m.def("InterpolationLayer_gpu_backward", &InterpolationLayer_gpu_backward, "Interpolation backward (HIP, gfx950)");
}
PYBIND11_MODULE(interpolationch_cuda, m) {
    m.def("InterpolationChLayer_gpu_forward", &InterpolationChLayer_gpu_forward, "InterpolationCh forward (HIP, gfx950)");
    m.def("InterpolationChLayer_gpu_backward", &InterpolationChLayer_gpu_backward, "InterpolationCh backward (HIP, gfx950)");
}
PYBIND11_MODULE(separableconv_cuda, m) {
    m.def("SeparableConvLayer_gpu_forward", &SeparableConvLayer_gpu_forward, "SeparableConv forward (HIP, gfx950)");
    m.def("SeparableConvLayer_gpu_backward", &SeparableConvLayer_gpu_backward, "SeparableConv backward (HIP, gfx950)");
}
PYBIND11_MODULE(separableconvflow_cuda, m) {
    m.def("SeparableConvFlowLayer_gpu_forward", &SeparableConvFlowLayer_gpu_forward, "SeparableConvFlow forward (HIP, gfx950)");
    m.def("SeparableConvFlowLayer_gpu_backward", &SeparableConvFlowLayer_gpu_backward, "SeparableConvFlow backward (HIP, gfx950)");
}
PYBIND11_MODULE(correlation_cuda, m) {
    m.def("forward", &correlation_forward, "Correlation forward (HIP, gfx950)");
    m.def("backward", &correlation_backward, "Correlation backward (HIP, gfx950)");
}
