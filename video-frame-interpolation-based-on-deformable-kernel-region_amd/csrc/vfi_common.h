// vfi_common.h -- shared device/host helpers for the gfx950 kernels of libvfi_hip.so.
//
// Numerics contract (DESIGN.md "numerics"): the library is compiled with
// -ffp-contract=off; every fused multiply-add is written explicitly with
// fmaf() at the positions where nvcc's default -fmad=true would fuse the
// reference's `acc += a*b` statements.  The CPU oracle's fmad=1 mode performs
// the same operations in the same order, so deterministic ops compare bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vfi_hip.h"

#define VFI_WAVE 64

// Pixel tile of the one-thread-per-pixel kernels: one wave = one 64-pixel row
// segment (256-B coalesced rows of every plane), four rows per workgroup.
#define VFI_TX 64
#define VFI_TY 4

namespace vfi {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// (1-a)(1-b)*TL + a(1-b)*TR + (1-a)b*BL + ab*BR, left to right, adds fused
// (filterinterpolation_cuda_kernel.cu:2789-2793; interpolation_cuda_kernel.cu:86-87)
__device__ __forceinline__ float blend4(float a, float b, float TL, float TR, float BL, float BR) {
    const float w00 = (1.0f - a) * (1.0f - b);
    const float w10 = a * (1.0f - b);
    const float w01 = (1.0f - a) * b;
    const float w11 = a * b;
    float t = w00 * TL;
    t = fmaf(w10, TR, t);
    t = fmaf(w01, BL, t);
    t = fmaf(w11, BR, t);
    return t;
}

// validity test of the adaptive-warping layer (filterinterpolation_cuda_kernel.cu:2735-2736)
__device__ __forceinline__ bool fi_valid(float fx, float fy, float x2, float y2, int w, int h) {
    return x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(w - 1) && y2 <= (float)(h - 1) &&
           fabsf(fx) < (float)w / 2.0f && fabsf(fy) < (float)h / 2.0f;
}

// Wave-wide min / max as six DPP steps at VALU rate (prefix within each row of 16 lanes, then
// row_bcast:15 and row_bcast:31 carry the row results to lane 63), read back as a scalar.  The
// __shfl_xor butterfly compiles to six dependent ds_bpermute_b32, each an LDS round trip.
#define VFI_DPP_STEP(OP, CTRL, ROWMASK) v = OP(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ int wave_min_i32(int v) {
    VFI_DPP_STEP(min, 0x111, 0xf); VFI_DPP_STEP(min, 0x112, 0xf); VFI_DPP_STEP(min, 0x114, 0xf);
    VFI_DPP_STEP(min, 0x118, 0xf); VFI_DPP_STEP(min, 0x142, 0xa); VFI_DPP_STEP(min, 0x143, 0xc);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) {
    VFI_DPP_STEP(max, 0x111, 0xf); VFI_DPP_STEP(max, 0x112, 0xf); VFI_DPP_STEP(max, 0x114, 0xf);
    VFI_DPP_STEP(max, 0x118, 0xf); VFI_DPP_STEP(max, 0x142, 0xa); VFI_DPP_STEP(max, 0x143, 0xc);
    return __builtin_amdgcn_readlane(v, 63);
}
#undef VFI_DPP_STEP

inline dim3 pixel_grid(int w, int h, int batch) {
    return dim3((unsigned)((w + VFI_TX - 1) / VFI_TX), (unsigned)((h + VFI_TY - 1) / VFI_TY), (unsigned)batch);
}

// Development knobs: tunables that experiments (tools/) flip at run time exist only in a -DVFI_DEV build of the
// library (make OUT=../lib_dev EXTRA=-DVFI_DEV); in the product they are compile-time constants and no setter is
// exported, so no caller or thread can change kernel selection for another.
#ifdef VFI_DEV
#define VFI_KNOB(type, name, value) static type name = value
#else
#define VFI_KNOB(type, name, value) static constexpr type name = value
#endif

// compute units of the CURRENT device (cached per device id)
int device_cu_count();
// How many channel groups (blockIdx.y) the staged FilterInterpolation kernels split a tile's channel range into.
// Cost of g groups, in channels: a workgroup pays `prologue` channels' worth before its first channel (flow, filter, bounding
// box, first window); a launch leaves half a round of its slots (2 workgroups per CU) idle at the end on average; and the
// more workgroups a slot runs, the better their unequal durations even out (they go to whichever slot frees first).
// Fitted to launches timed in isolation (tools/fi_isolated.py; fs=4, 1080p, C=196: 1 group 1.12 ms, 2 1.00-1.03, 3 0.99,
// 4 1.00, 8 1.08).
int fi_channel_groups(int ntiles, int channel, double prologue);

// Deterministic image gradients.  The reference scatters the image gradient of its warping layers with fp32 atomics
// (filterinterpolation_cuda_kernel.cu:2890-2942, interpolation_cuda_kernel.cu:154-157): the sum depends on the order the
// atomics arrive in, so two runs differ in the last bits.  Here every addend is scaled by ONE power of two per call,
// rounded to an integer and added with a 64-bit INTEGER atomic into a scratch plane; a last pass converts the exact
// integer sums to float once and adds them to the caller's (zero-filled) gradient.  Order-free, hence reproducible bit
// for bit.  The scale is 2^(62 - ceil(log2(h w T)) - eg - ew) with 2^eg > max |gradoutput|, 2^ew > max |tap weight|
// (the filter tensor; 1 where the weights are bilinear fractions only) and T = the taps of a pixel (fs x fs; 4 for a
// bilinear sample): an addend is below 2^(62 - ceil(log2(h w T))) and even a cell that EVERY tap of EVERY pixel of the frame
// hits (border clamping folds a pixel's taps onto one cell) stays inside 63 bits -- no combination of finite inputs overflows.
// Non-finite inputs: the first pass raises a flag when gradoutput or the weights hold a NaN or an infinity, and the
// kernels then scatter with the reference's own fp32 atomics for that call (NaN / Inf propagate to exactly the cells
// the reference would poison; the integer path would turn them into finite garbage).
//   host:   gradacc_begin (zeroes the scratch; largest |gradoutput|, largest |weight|, non-finite flag)  ->  the backward kernel
//           ->  gradacc_finish
//   device: gradacc_ctx(hdr) once per thread, gradacc_add(...) per addend; cells are indexed densely [b][c][y][x] whatever
//           the strides of the gradient tensor.
// hdr words: [0] bits of max |gradoutput|, [1] non-finite flag, [2] bits of max |weight| (0: none given), [3] ceil(log2(h w T))
struct GradAccCtx { float scale; bool nonfinite; };
__device__ __forceinline__ int gradacc_exponent(const int* __restrict__ hdr) {
    int eg = 0, ew = 1;                                     // no weight tensor: |weight| <= 1 < 2^1
    (void)frexpf(__int_as_float(hdr[0]), &eg);
    if (hdr[2] != 0) (void)frexpf(__int_as_float(hdr[2]), &ew);
    return max(-126, min(126, 62 - hdr[3] - eg - max(ew, 1)));
}
__device__ __forceinline__ GradAccCtx gradacc_ctx(const int* __restrict__ hdr) {
    GradAccCtx c;
    c.scale = ldexpf(1.0f, gradacc_exponent(hdr));
    c.nonfinite = hdr[1] != 0;
    return c;
}
// acc_plane / g_plane: the channel's plane of the dense scratch and of the caller's gradient tensor; di / gi: the cell's
// index in each
__device__ __forceinline__ void gradacc_add(unsigned long long* acc_plane, float* g_plane, int64_t di, int64_t gi, float v,
                                            const GradAccCtx& cx) {
    if (cx.nonfinite) atomicAdd(&g_plane[gi], v);
    else atomicAdd(&acc_plane[di], (unsigned long long)__float2ll_rn(v * cx.scale));
}
// weights (may be null): a [batch, wchannel, h, w] tensor whose largest |element| bounds the tap weights; wchannel = the taps
// of a pixel (also when weights is null; below 4: 4);
// nflags / flags: that many zeroed words of the same scratch for the caller's kernels (flags may be null)
int gradacc_begin(hipStream_t st, const float* gout, int batch, int channel, int h, int w, vfi_strides sg,
                  const float* weights, int wchannel, vfi_strides sw, unsigned long long** acc, int** hdr,
                  int nflags = 0, int** flags = nullptr);
int gradacc_finish(hipStream_t st, const unsigned long long* acc, const int* hdr, float* g1, int batch, int channel, int h, int w,
                   vfi_strides s1);

inline int launch_status() {
    return hipGetLastError() == hipSuccess ? VFI_OK : VFI_ERR_LAUNCH;
}

}  // namespace vfi
