/*
 * vfi_hip.h -- C ABI of libvfi_hip.so, the MI355X (gfx950) implementation of the
 * DAIN / VFIDKR frame-synthesis hot path.
 *
 * One entry point per function the reference's pybind11 extension modules
 * export.  No torch types cross this boundary: device
 * pointers, sizes, element strides and a HIP stream.  The torch extension
 * modules in csrc/shim/ (same module and function names as the reference's)
 * and the ctypes loader in the package are both thin callers of this ABI.
 *
 * Conventions (kept from the reference bindings unless stated):
 *  - tensors are float32, NCHW, innermost (w) stride 1; `*_s` arguments are the
 *    batch / channel / row strides IN ELEMENTS (int64: a 196-channel 4K tensor
 *    overflows the reference's 32-bit offsets for batch >= 2);
 *  - all pointers are DEVICE pointers valid on the device `stream` belongs to;
 *  - calls are asynchronous: kernels are enqueued on `stream`, nothing syncs;
 *  - return value: 0 = success, 1 = shape/stride problem (the reference
 *    bindings' silent `return 1`), VFI_ERR_LAUNCH = a HIP launch failed (the
 *    reference bindings raise AT_ERROR("CUDA call failed") for it);
 *  - the reference's my_package ops expect the CALLER to zero-fill outputs /
 *    counts / grads (FilterInterpolationLayer.py:34, FlowProjectionLayer.py:35-36).
 *    Every FORWARD entry point here writes every element of its outputs
 *    (`count` and `output` of the projections included), so forward outputs need
 *    no zero fill; a zero-filled buffer is of course accepted.  Every grad*
 *    buffer of a BACKWARD must arrive zeroed exactly as in the reference.
 */
#ifndef VFI_HIP_H
#define VFI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFI_OK          0
#define VFI_ERR_SHAPE   1
#define VFI_ERR_LAUNCH  (-2)

typedef void* vfi_stream_t;           /* hipStream_t */

/* strides of one NCHW tensor, in elements; w stride is 1 by contract */
typedef struct vfi_strides {
    int64_t b, c, h;
} vfi_strides;

/* library / build identification: "vfi_hip <version> gfx950" */
const char* vfi_version(void);

/* ---- filterinterpolation_cuda ------------------------------------------------
 * replaces FilterInterpolationLayer_gpu_forward_ori / _backward_ori
 * (filterinterpolation_cuda.cc:537-606, 608-687).  filter_channels = input3.size(1);
 * filter_size = (int)sqrt((float)filter_channels) as the binding computes it.
 * output uses input1's strides (the binding checks they are equal, cc:582-583). */
int vfi_filterinterp_forward_ori(const float* input1, const float* input2, const float* input3,
                                 float* output,
                                 int batch, int channel, int h, int w, int filter_channels,
                                 vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                 vfi_stream_t stream);
int vfi_filterinterp_backward_ori(const float* input1, const float* input2, const float* input3,
                                  const float* gradoutput,
                                  float* gradinput1, float* gradinput2, float* gradinput3,
                                  int batch, int channel, int h, int w, int filter_channels,
                                  vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                  vfi_stream_t stream);

/* The same forward for SEVERAL flows over one image and one filter: outputs[t] = the function above with
 * input2 = flows[t], t < nflows (bit for bit).  This is what DAIN_slowmotion does per direction: FilterInterpolate_ctx
 * (networks/DAIN_slowmotion.py:167-183, 311-317) warps the same context tensor with the same filter once per time
 * offset.  With filter_channels == 16 one launch stages ONE window per tile and channel for two flows (half the image
 * traffic per output; more flows go in pairs, a last odd one alone: the three time offsets of DAIN_slowmotion x4 = one
 * two-flow launch + one single-flow launch -- measured faster than a three-flow launch, whose union window outgrows the
 * LDS ring); other filter sizes are single launches.
 * flows / outputs: HOST arrays of nflows device pointers; every flow has strides s2, every output s1. */
int vfi_filterinterp_forward_ori_multi(const float* input1, const float* const* flows, const float* input3,
                                       float* const* outputs, int nflows,
                                       int batch, int channel, int h, int w, int filter_channels,
                                       vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                       vfi_stream_t stream);

/* fp16 STORAGE, fp32 arithmetic (BASELINE.json configs[2]): input1 and output are IEEE
 * half tensors (strides in half elements), flow and filter stay float32.  The result is the fp32
 * result of the function above on the widened inputs, rounded to half once; the LDS-staged path
 * (filter_channels == 16) sums the 16 products in one chain and agrees with that to fp16 rounding. */
int vfi_filterinterp_forward_ori_f16(const void* input1_half, const float* input2, const float* input3,
                                     void* output_half,
                                     int batch, int channel, int h, int w, int filter_channels,
                                     vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                     vfi_stream_t stream);

/* deformable-kernel variants (filterinterpolation_cuda.cc:11-92, 191-272, 374-447).
 * variant: 0 = FilterInterpolationLayer_gpu_forward (4 inputs, fs in {4,6}),
 *          1 = ..._forward_deforconv, 2 = ..._forward_nofilterwithdeforconv
 *              (input3 is the 2*fs*fs offset field, input4 unused/NULL).
 * Documented divergence: the reference reads out of bounds when a displaced tap
 * leaves the image; here the four bilinear corners are clamped to the image. */
#define VFI_DEFOR_OFFSET   0
#define VFI_DEFOR_REGION   1
#define VFI_DEFOR_NOFILTER 2
int vfi_filterinterp_forward_defor(int variant,
                                   const float* input1, const float* input2, const float* input3,
                                   const float* input4, float* output,
                                   int batch, int channel, int h, int w, int filter_size,
                                   vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4,
                                   vfi_stream_t stream);

/* backward of the same three variants: FilterInterpolationLayer_gpu_backward (cc:93-187),
 * ..._backward_deforconv (cc:273-367), ..._backward_nofilterwithdeforconv (cc:448-533).
 * gradinput1..4 must arrive zeroed.  variant 2: input3 / gradinput3 are the offset field and its
 * gradient, input4 / gradinput4 unused (NULL).  gradoutput is addressed with input1's strides,
 * gradinput_k with input_k's, as in the reference. */
int vfi_filterinterp_backward_defor(int variant,
                                    const float* input1, const float* input2, const float* input3,
                                    const float* input4, const float* gradoutput,
                                    float* gradinput1, float* gradinput2, float* gradinput3, float* gradinput4,
                                    int batch, int channel, int h, int w, int filter_size,
                                    vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4,
                                    vfi_stream_t stream);

/* ---- flowprojection_cuda -----------------------------------------------------
 * replaces FlowProjectionLayer_gpu_forward / _backward (flowprojection_cuda.cc:9-57, 59-114).
 * count [B,1,H,W] and output [B,2,H,W] are fully written (no zero fill needed).
 * The forward keeps a per-(device, stream) workspace (block tables, bitmaps, three scratch planes: about
 * 13 bytes per pixel), allocated on the first call for a stream and replaced by a larger one when a larger
 * frame arrives.  hipMalloc is not legal inside a stream capture: call vfi_projection_reserve (or make one
 * warm-up call) before capturing into a HIP graph.  A buffer that has been outgrown is retired, never freed,
 * so a graph captured earlier stays replayable; vfi_release_workspaces() frees everything. */
int vfi_flowprojection_forward(const float* input1, float* count, float* output,
                               int batch, int h, int w, int fillhole,
                               vfi_strides s1, vfi_strides sc,
                               vfi_stream_t stream);
int vfi_flowprojection_backward(const float* input1, const float* count, const float* gradoutput,
                                float* gradinput1,
                                int batch, int h, int w,
                                vfi_strides s1, vfi_strides sc,
                                vfi_stream_t stream);

/* FlowProject(inputs, depth) of the networks (networks/DAIN.py:533-539, networks/DAIN_slowmotion.py:301-307 -- the reference
 * loops over the list, one FlowProjectionModule call per flow, and does so for both directions back to back,
 * DAIN.py:215-220 / DAIN_slowmotion.py:156-159): the whole list in ONE launch triple (per 8 items), so that the three
 * dependent launches of a projection are paid once per list, not once per flow.  inputs1 / counts / outputs: HOST arrays of
 * nitems device pointers, item i = [batch,2,h,w] flow -> [batch,1,h,w] count + [batch,2,h,w] output; all items share
 * shape and strides (s1: flows and outputs, sc: counts).  Every item needs its own count and output (they are written
 * concurrently: two equal pointers are refused with VFI_ERR_SHAPE).  Results per item: vfi_flowprojection_forward's, bit for
 * bit.  The workspace must cover nitems * batch frames: vfi_projection_reserve(nitems * batch, h, w, stream) before a capture. */
int vfi_flowprojection_forward_batch(const float* const* inputs1, float* const* counts, float* const* outputs,
                                     int nitems, int batch, int h, int w, int fillhole,
                                     vfi_strides s1, vfi_strides sc,
                                     vfi_stream_t stream);

/* Make the projection workspace of `stream` large enough for [batch, *, h, w] frames (both projections and
 * their _up4 forms; h, w are the full-resolution sizes).  Allocates, so call it outside a capture. */
int vfi_projection_reserve(int batch, int h, int w, vfi_stream_t stream);
/* Synchronises the devices involved and frees every workspace the library holds (live and retired) on every
 * stream.  Call only when no launch or captured graph of this library is still to run; later calls
 * allocate afresh. */
int vfi_release_workspaces(void);

/* ---- depthflowprojection_cuda ------------------------------------------------
 * replaces DepthFlowProjectionLayer_gpu_forward / _backward
 * (depthflowprojection_cuda.cc:10-68, 70-139). */
int vfi_depthflowprojection_forward(const float* input1, const float* input2,
                                    float* count, float* output,
                                    int batch, int h, int w, int fillhole,
                                    vfi_strides s1, vfi_strides s2, vfi_strides sc,
                                    vfi_stream_t stream);
/* the list form (see vfi_flowprojection_forward_batch); inputs2[i] is item i's depth weight [batch,1,h,w] with strides s2 --
 * items may share one (DAIN_slowmotion projects every time offset of a direction with that direction's depth) */
int vfi_depthflowprojection_forward_batch(const float* const* inputs1, const float* const* inputs2,
                                          float* const* counts, float* const* outputs,
                                          int nitems, int batch, int h, int w, int fillhole,
                                          vfi_strides s1, vfi_strides s2, vfi_strides sc,
                                          vfi_stream_t stream);
int vfi_depthflowprojection_backward(const float* input1, const float* input2,
                                     const float* count, const float* output,
                                     const float* gradoutput,
                                     float* gradinput1, float* gradinput2,
                                     int batch, int h, int w,
                                     vfi_strides s1, vfi_strides s2, vfi_strides sc,
                                     vfi_stream_t stream);

/* ---- mindepthflowprojection_cuda --------------------------------------------
 * replaces minDepthFlowProjectionLayer_gpu_forward / _backward
 * (mindepthflowprojection_cuda.cc:12-66, 68-139).  Each target keeps the (negated) flow of the
 * source with the largest weight `input2` that lands on it (top-left integer neighbour only) and
 * that weight in `count`.  The reference does this with an unguarded read-compare-write and is
 * scheduling dependent; this library returns what those statements give when the sources are
 * visited in raster order (largest weight above the incoming `count`, first source on ties).
 * count and output must be zero-filled by the caller (untouched targets keep their values).
 * The forward keeps a per-stream scratch buffer of about 8.3 bytes per pixel (keys + two bitmaps).  The backward gives
 * gradinput1 only: the reference never writes gradinput2 (its code for it is commented out). */
int vfi_mindepthflowprojection_forward(const float* input1, const float* input2,
                                       float* count, float* output,
                                       int batch, int h, int w, int fillhole,
                                       vfi_strides s1, vfi_strides s2, vfi_strides sc,
                                       vfi_stream_t stream);
int vfi_mindepthflowprojection_backward(const float* input1, const float* input2,
                                        const float* count, const float* gradoutput,
                                        float* gradinput1,
                                        int batch, int h, int w,
                                        vfi_strides s1, vfi_strides s2, vfi_strides sc,
                                        vfi_stream_t stream);

/* ---- interpolation_cuda / interpolationch_cuda -------------------------------
 * replaces Interpolation[Ch]Layer_gpu_forward / _backward (interpolation_cuda.cc:10-60,
 * 63-121).  The C==3 restriction of `interpolation_cuda` lives in its shim. */
int vfi_interpolation_forward(const float* input1, const float* input2, float* output,
                              int batch, int channel, int h, int w,
                              vfi_strides s1, vfi_strides s2,
                              vfi_stream_t stream);
int vfi_interpolation_backward(const float* input1, const float* input2, const float* gradoutput,
                               float* gradinput1, float* gradinput2,
                               int batch, int channel, int h, int w,
                               vfi_strides s1, vfi_strides s2,
                               vfi_stream_t stream);

/* ---- separableconv_cuda ------------------------------------------------------
 * replaces SeparableConvLayer_gpu_forward / _backward (separableconv_cuda.cc:10-87, 88-174).
 * h, w are input1's; input2/input3/output are [B, fs | C, h-fs+1, w-fs+1]. */
int vfi_separableconv_forward(const float* input1, const float* input2, const float* input3,
                              float* output,
                              int batch, int channel, int h, int w, int filter_size,
                              vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides so,
                              vfi_stream_t stream);
int vfi_separableconv_backward(const float* input1, const float* input2, const float* input3,
                               const float* gradoutput,
                               float* gradinput1, float* gradinput2, float* gradinput3,
                               int batch, int channel, int h, int w, int filter_size,
                               vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides so,
                               vfi_stream_t stream);

/* ---- separableconvflow_cuda --------------------------------------------------
 * replaces SeparableConvFlowLayer_gpu_forward / _backward (separableconvflow_cuda.cc:9-102,
 * 103-199).  input1 is only shape-checked by the reference and is not passed. */
int vfi_separableconvflow_forward(const float* input2, const float* input3, float* flow_output,
                                  int batch, int h, int w, int filter_size,
                                  vfi_strides s2, vfi_strides s3, vfi_strides so,
                                  vfi_stream_t stream);
int vfi_separableconvflow_backward(const float* input2, const float* input3,
                                   const float* gradflow_output,
                                   float* gradinput2, float* gradinput3,
                                   int batch, int h, int w, int filter_size,
                                   vfi_strides s2, vfi_strides s3, vfi_strides so,
                                   vfi_stream_t stream);

/* ---- correlation_cuda --------------------------------------------------------
 * replaces correlation_cuda.forward / .backward (correlation_cuda.cc:8-85, 87-165).
 * Inputs and outputs are DENSE NCHW (the reference kernels ignore strides).
 * The padded NHWC repack buffers rInput1/rInput2 of the reference are not
 * needed by these kernels; the shim still resizes and zero-fills them so a
 * caller that inspects them sees the reference's shapes.
 * vfi_correlation_output_dims reproduces correlation_cuda.cc:23-36. */
int vfi_correlation_output_dims(int h, int w, int pad_size, int kernel_size, int max_displacement,
                                int stride1, int stride2,
                                int* out_channels, int* out_h, int* out_w);
int vfi_correlation_forward(const float* input1, const float* input2, float* output,
                            int batch, int channel, int h, int w,
                            int pad_size, int kernel_size, int max_displacement,
                            int stride1, int stride2,
                            vfi_stream_t stream);
/* Two forward calls of equal shape in ONE launch: the same pyramid level of the two flow networks a frame pair runs,
 * (I0, I1) and (I1, I0) (networks/DAIN.py:196-202; PWCNet/PWCNet.py:230-300 calls the layer once per level).  At the coarse
 * levels a launch is pure latency (13-18 us for 1 MB at 1080p) and two cost what one does.  Results: the two
 * vfi_correlation_forward calls', bit for bit.  (Callers that batch the two orders along dim 0 need nothing new: batch = 2.) */
int vfi_correlation_forward_pair(const float* input1_a, const float* input2_a, float* output_a,
                                 const float* input1_b, const float* input2_b, float* output_b,
                                 int batch, int channel, int h, int w,
                                 int pad_size, int kernel_size, int max_displacement,
                                 int stride1, int stride2,
                                 vfi_stream_t stream);
/* The reference's at::Half instantiation of the forward (correlation_cuda_kernel.cu:386, 403): inputs and output
 * IEEE half, each product rounded to half, float accumulation, mean rounded to half once. */
int vfi_correlation_forward_f16(const void* input1_half, const void* input2_half, void* output_half,
                                int batch, int channel, int h, int w,
                                int pad_size, int kernel_size, int max_displacement,
                                int stride1, int stride2,
                                vfi_stream_t stream);
int vfi_correlation_backward(const float* input1, const float* input2, const float* gradoutput,
                             float* gradinput1, float* gradinput2,
                             int batch, int channel, int h, int w,
                             int pad_size, int kernel_size, int max_displacement,
                             int stride1, int stride2,
                             vfi_stream_t stream);

/* ==== glue either side of the ops above: the reference does these with torch
 * built-ins and Python; here each is one launch.  No reference binding exists for them: the
 * host-side mirrors are <pkg>/fused.py. ========================================================= */

/* forward_flownets (networks/DAIN_slowmotion.py:204-216, DAIN.py:296-311):
 * output[B,C,4hq,4wq] = nn.Upsample(scale_factor=4, mode='bilinear')((mul0 * input) * mul1),
 * torch's align_corners=False rule.  mul0 = div_flow, mul1 = the time offset. */
int vfi_flow_upsample4(const float* input, float* output,
                       int batch, int channels, int hq, int wq, float mul0, float mul1,
                       vfi_strides sq, vfi_strides so,
                       vfi_stream_t stream);

/* forward_flownets + FlowProject in one call: the quarter-resolution flow [B,2,hq,wq] is upsampled
 * into a per-stream scratch tensor of the library and projected; count [B,1,4hq,4wq] and output
 * [B,2,4hq,4wq] as vfi_flowprojection_forward.  Results equal vfi_flow_upsample4 followed by
 * vfi_[depth]flowprojection_forward bit for bit (it is those two steps without a caller-side
 * tensor). */
int vfi_flowprojection_forward_up4(const float* flow_q, float* count, float* output,
                                   int batch, int hq, int wq, float mul0, float mul1, int fillhole,
                                   vfi_strides sq, vfi_strides sc, vfi_strides so,
                                   vfi_stream_t stream);
int vfi_depthflowprojection_forward_up4(const float* flow_q, const float* input2,
                                        float* count, float* output,
                                        int batch, int hq, int wq, float mul0, float mul1, int fillhole,
                                        vfi_strides sq, vfi_strides s2, vfi_strides sc, vfi_strides so,
                                        vfi_stream_t stream);

/* FilterInterpolate (networks/DAIN_slowmotion.py:324-335, DAIN.py:560-573): out0 = A1(ref0, flow0,
 * filt0), out2 = A1(ref2, flow2, filt2), blend = out0 * w0 + out2 * w2 (products rounded
 * separately, as torch's three elementwise ops).  out0 / out2 may be NULL.  ref0/ref2, flow0/flow2
 * and filt0/filt2 share strides pairwise; blend/out0/out2 share s_out. */
int vfi_filterinterp_blend_forward(const float* ref0, const float* ref2,
                                   const float* flow0, const float* flow2,
                                   const float* filt0, const float* filt2,
                                   float* blend, float* out0, float* out2,
                                   int batch, int channel, int h, int w, int filter_channels,
                                   float w0, float w2,
                                   vfi_strides s_ref, vfi_strides s_flow, vfi_strides s_filt, vfi_strides s_out,
                                   vfi_stream_t stream);

/* PWCDCNet.warp (PWCNet/PWCNet.py:159-199): output = grid_sample(x, grid(flow)) * mask, mask = 1
 * where grid_sample(ones, grid) >= 0.9999 else 0; bilinear, zeros padding.  align_corners: 1 = the
 * grid_sample of torch <= 1.2 the reference was written for, 0 = the default of torch >= 1.3. */
int vfi_pwc_warp_forward(const float* x, const float* flow, float* output,
                         int batch, int channel, int h, int w, int align_corners,
                         vfi_strides sx, vfi_strides sf, vfi_strides so,
                         vfi_stream_t stream);

/* PWCDCNet's warp feeding its correlation layer (PWCNet/PWCNet.py:244-247, 266-267, 282-283, 299-300):
 * output[B,81,h,w] = correlation(input1, warp(input2, flow)) with pad 4, kernel 1, max displacement 4, strides 1 --
 * vfi_pwc_warp_forward followed by vfi_correlation_forward, bit for bit, in one launch and without the warped
 * tensor.  input1 / input2 / output dense NCHW, flow [B,2,h,w] with strides sf. */
int vfi_pwc_warp_correlation_forward(const float* input1, const float* input2, const float* flow, float* output,
                                     int batch, int channel, int h, int w, int align_corners,
                                     vfi_strides sf, vfi_stream_t stream);

/* frame boundary (demo_MiddleBury.py:280-318, 350-364, 370-388).
 * u8 -> planar: dst[b,c,y,x] = src[b, clamp(y - pad_top), clamp(x - pad_left), c] / 255 for the padded
 * frame (h + pad_top + pad_bottom) x (w + pad_left + pad_right); src is dense [B,h,w,3] uint8.
 * planar -> u8: dst[b,y,x,c] = uint8(rint(255 * clip(src[b,c,top+y,left+x], 0, 1))), dst dense [B,h,w,3].
 * error sums: sums[0] += sum|a-b|, sums[1] += sum (a-b)^2 over n bytes (exact; caller zeroes sums).
 * ssim sums: sums[0] += sum of the SSIM map of every colour plane of every frame pair as the demo computes it
 *   (demo_MiddleBury.py:40-162, 382-388: planes / 255, 11-tap sigma-1.5 Gaussian along H then W without padding,
 *   data_range 1, K = (0.01, 0.03)), each value as a 2^-32 fixed-point integer (order-free; caller zeroes sums).
 *   mean SSIM = sums[0] / 2^32 / (batch * 3 * (h - 10) * (w - 10)); a dimension below 11 is not smoothed and
 *   contributes its full length instead of (length - 10), as in the reference.  a, b: dense [B,h,w,3] uint8. */
int vfi_frame_u8_to_planar(const unsigned char* src_hwc, float* dst,
                           int batch, int h, int w, int pad_left, int pad_right, int pad_top, int pad_bottom,
                           vfi_strides sd, vfi_stream_t stream);
int vfi_planar_to_frame_u8(const float* src, unsigned char* dst_hwc,
                           int batch, int h, int w, int top, int left,
                           vfi_strides ss, vfi_stream_t stream);
int vfi_frame_error_sums(const unsigned char* a, const unsigned char* b, int64_t n,
                         unsigned long long* sums, vfi_stream_t stream);
int vfi_frame_ssim_sums(const unsigned char* a, const unsigned char* b, int batch, int h, int w,
                        long long* sums, vfi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VFI_HIP_H */
