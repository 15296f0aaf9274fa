// projection.hip -- FlowProjection / DepthFlowProjection (forward splat of the
// t->0 flow to the intermediate time) for gfx950.
//
// Semantics: flowprojection_cuda_kernel.cu:29-301 and
// depthflowprojection_cuda_kernel.cu:29-341 of the reference; entry points
// replace flowprojection_cuda.cc / depthflowprojection_cuda.cc.
//
// Forward = three passes on one stream, as in the reference (the hole filler
// needs the complete count plane):
//   1. splat: every source pixel adds (-d*fx, -d*fy, d) to its 4 integer
//      neighbours (d = 1 without depth);
//   2. normalise where count > 0;
//   3. optional hole fill from the nearest non-hole in -x, +x, -y, +y.
#include "vfi_common.h"

namespace vfi {

template <bool DEPTH>
__global__ __launch_bounds__(VFI_TX * VFI_TY) void proj_splat(
    const float* __restrict__ in1, const float* __restrict__ in2, float* count, float* out,
    int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides sc) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* flow = in1 + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    const float fx = flow[0];
    const float fy = flow[s1.c];
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(w - 1) && y2 <= (float)(h - 1))) return;
    const int L = (int)x2, T = (int)y2;
    const int R = min(L + 1, w - 1), Bm = min(T + 1, h - 1);
    float ax = -fx, ay = -fy, ac = 1.0f;
    if constexpr (DEPTH) {
        const float d = in2[(int64_t)b * s2.b + (int64_t)y * s2.h + x];
        ax = -d * fx; ay = -d * fy; ac = d;
    }
    float* o0 = out + (int64_t)b * s1.b;
    float* o1 = o0 + s1.c;
    float* cn = count + (int64_t)b * sc.b;
    const int64_t oT = (int64_t)T * s1.h, oB = (int64_t)Bm * s1.h;
    const int64_t cT = (int64_t)T * sc.h, cB = (int64_t)Bm * sc.h;
    // R == L / Bm == T at the far edges: the same cell receives the value twice (:72-73)
    atomicAdd(&o0[oT + L], ax); atomicAdd(&o0[oT + R], ax); atomicAdd(&o0[oB + L], ax); atomicAdd(&o0[oB + R], ax);
    atomicAdd(&o1[oT + L], ay); atomicAdd(&o1[oT + R], ay); atomicAdd(&o1[oB + L], ay); atomicAdd(&o1[oB + R], ay);
    atomicAdd(&cn[cT + L], ac); atomicAdd(&cn[cT + R], ac); atomicAdd(&cn[cB + L], ac); atomicAdd(&cn[cB + R], ac);
}

__global__ __launch_bounds__(VFI_TX * VFI_TY) void proj_average(
    const float* __restrict__ count, float* out, int h, int w, vfi_strides s1, vfi_strides sc) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float c = count[(int64_t)b * sc.b + (int64_t)y * sc.h + x];
    if (c > 0.0f) {
        float* o = out + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
        o[0] /= c;
        o[s1.c] /= c;
    }
}

// pass 3 (flowprojection_cuda_kernel.cu:175-231).  A cell read here is either a
// non-hole (never written by this pass) or is multiplied by 0.
__global__ __launch_bounds__(VFI_TX * VFI_TY) void proj_fillhole(
    const float* __restrict__ count, float* out, int h, int w, vfi_strides s1, vfi_strides sc) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* cn = count + (int64_t)b * sc.b;
    if (!(cn[(int64_t)y * sc.h + x] <= 0.0f)) return;
    int lo = x; float lt = 0.0f;
    while (lt == 0.0f && lo - 1 >= 0) { lo -= 1; lt = cn[(int64_t)y * sc.h + lo]; }
    int ro = x; float rt = 0.0f;
    while (rt == 0.0f && ro + 1 <= w - 1) { ro += 1; rt = cn[(int64_t)y * sc.h + ro]; }
    int uo = y; float ut = 0.0f;
    while (ut == 0.0f && uo - 1 >= 0) { uo -= 1; ut = cn[(int64_t)uo * sc.h + x]; }
    int dn = y; float dt = 0.0f;
    while (dt == 0.0f && dn + 1 <= h - 1) { dn += 1; dt = cn[(int64_t)dn * sc.h + x]; }
    if (lt + rt + ut + dt <= 0.0f) return;
    lt = (lt > 0.0f) ? 1.0f : 0.0f;
    rt = (rt > 0.0f) ? 1.0f : 0.0f;
    ut = (ut > 0.0f) ? 1.0f : 0.0f;
    dt = (dt > 0.0f) ? 1.0f : 0.0f;
    const float den = lt + rt + ut + dt;
    float* o0 = out + (int64_t)b * s1.b;
    float* o1 = o0 + s1.c;
    const int64_t row = (int64_t)y * s1.h;
    o0[row + x] = (lt * o0[row + lo] + rt * o0[row + ro] + ut * o0[(int64_t)uo * s1.h + x] +
                   dt * o0[(int64_t)dn * s1.h + x]) / den;
    o1[row + x] = (lt * o1[row + lo] + rt * o1[row + ro] + ut * o1[(int64_t)uo * s1.h + x] +
                   dt * o1[(int64_t)dn * s1.h + x]) / den;
}

template <bool DEPTH>
__global__ __launch_bounds__(VFI_TX * VFI_TY) void proj_backward(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ count,
    const float* __restrict__ fwd_out, const float* __restrict__ gout, float* g1, float* g2,
    int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides sc) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* flow = in1 + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    const float fx = flow[0];
    const float fy = flow[s1.c];
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(w - 1) && y2 <= (float)(h - 1))) return;
    const int L = (int)x2, T = (int)y2;
    const int R = min(L + 1, w - 1), Bm = min(T + 1, h - 1);
    const int64_t to[4] = { (int64_t)T * s1.h + L, (int64_t)T * s1.h + R, (int64_t)Bm * s1.h + L, (int64_t)Bm * s1.h + R };
    const int64_t tc[4] = { (int64_t)T * sc.h + L, (int64_t)T * sc.h + R, (int64_t)Bm * sc.h + L, (int64_t)Bm * sc.h + R };
    const float* cn = count + (int64_t)b * sc.b;
    const float* go = gout + (int64_t)b * s1.b;
    float* g = g1 + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    float d = 1.0f;
    if constexpr (DEPTH) d = in2[(int64_t)b * s2.b + (int64_t)y * s2.h + x];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        float acc = g[(int64_t)ch * s1.c];                  // caller zero-fills; accumulate as the reference does
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (DEPTH) acc += -go[(int64_t)ch * s1.c + to[k]] * d / cn[tc[k]];   // (:291-311)
            else                 acc += -go[(int64_t)ch * s1.c + to[k]] / cn[tc[k]];       // (:279-296)
        }
        g[(int64_t)ch * s1.c] = acc;
    }
    if constexpr (DEPTH) {
        const float* fo = fwd_out + (int64_t)b * s1.b;
        float* gd = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        float acc = gd[0];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float f = ch ? fy : fx;
#pragma unroll
            for (int k = 0; k < 4; ++k)                     // (:314-336)
                acc += -go[(int64_t)ch * s1.c + to[k]] / cn[tc[k]] * (f - fo[(int64_t)ch * s1.c + to[k]]);
        }
        gd[0] = acc;
    }
}

template <bool DEPTH>
static int project_forward(const float* in1, const float* in2, float* count, float* out, int batch, int h, int w,
                           int fillhole, vfi_strides s1, vfi_strides s2, vfi_strides sc, hipStream_t st) {
    const dim3 grid = pixel_grid(w, h, batch), block(VFI_TX, VFI_TY, 1);
    hipLaunchKernelGGL(proj_splat<DEPTH>, grid, block, 0, st, in1, in2, count, out, h, w, s1, s2, sc);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    hipLaunchKernelGGL(proj_average, grid, block, 0, st, count, out, h, w, s1, sc);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    if (fillhole) {
        hipLaunchKernelGGL(proj_fillhole, grid, block, 0, st, count, out, h, w, s1, sc);
        if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    }
    return VFI_OK;
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_flowprojection_forward(const float* input1, float* count, float* output, int batch, int h, int w,
                                           int fillhole, vfi_strides s1, vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !count || !output) return VFI_ERR_SHAPE;
    return project_forward<false>(input1, nullptr, count, output, batch, h, w, fillhole, s1, s1, sc, (hipStream_t)stream);
}

extern "C" int vfi_depthflowprojection_forward(const float* input1, const float* input2, float* count, float* output,
                                                int batch, int h, int w, int fillhole, vfi_strides s1, vfi_strides s2,
                                                vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !count || !output) return VFI_ERR_SHAPE;
    return project_forward<true>(input1, input2, count, output, batch, h, w, fillhole, s1, s2, sc, (hipStream_t)stream);
}

extern "C" int vfi_flowprojection_backward(const float* input1, const float* count, const float* gradoutput,
                                            float* gradinput1, int batch, int h, int w, vfi_strides s1, vfi_strides sc,
                                            vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !count || !gradoutput || !gradinput1) return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(proj_backward<false>, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input1, nullptr, count, nullptr, gradoutput, gradinput1, nullptr, h, w, s1, s1, sc);
    return launch_status();
}

extern "C" int vfi_depthflowprojection_backward(const float* input1, const float* input2, const float* count,
                                                 const float* output, const float* gradoutput, float* gradinput1,
                                                 float* gradinput2, int batch, int h, int w, vfi_strides s1,
                                                 vfi_strides s2, vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !count || !output || !gradoutput || !gradinput1 ||
        !gradinput2)
        return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(proj_backward<true>, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input1, input2, count, output, gradoutput, gradinput1, gradinput2, h, w, s1, s2, sc);
    return launch_status();
}
