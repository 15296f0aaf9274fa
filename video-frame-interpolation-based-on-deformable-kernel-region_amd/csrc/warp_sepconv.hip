// warp_sepconv.hip -- Interpolation / InterpolationCh (bilinear backward warp),
// SeparableConv (local separable convolution) and SeparableConvFlow (kernel
// centre of mass -> flow) for gfx950.
//
// Semantics: interpolation_cuda_kernel.cu:29-202, separableconv_cuda_kernel.cu:29-135,
// separableconvflow_cuda_kernel.cu:29-173 of the reference; entry points replace
// interpolation_cuda.cc, interpolationch_cuda.cc, separableconv_cuda.cc and
// separableconvflow_cuda.cc.  One thread per output pixel, a wave = 64
// consecutive x (coalesced plane rows), channel loop inside the thread.
#include "vfi_common.h"

namespace vfi {

// ------------------------------------------------------------------ Interpolation

__global__ __launch_bounds__(VFI_TX * VFI_TY) void interp_forward(
    const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out,
    int channel, int h, int w, vfi_strides s1, vfi_strides s2) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    const float fx = flow[0];
    const float fy = flow[s2.c];
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    float* dst = out + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    if (x2 >= 0.0f && y2 >= 0.0f && x2 < (float)w && y2 < (float)h) {       // strict upper bound (:71)
        const int L = (int)x2, T = (int)y2;
        const int R = min(L + 1, w - 1), Bm = min(T + 1, h - 1);
        const float alpha = x2 - (float)L, beta = y2 - (float)T;
        const float* img = in1 + (int64_t)b * s1.b;
        const int64_t oT = (int64_t)T * s1.h, oB = (int64_t)Bm * s1.h;
        for (int c = 0; c < channel; ++c) {
            const float* p = img + (int64_t)c * s1.c;
            dst[(int64_t)c * s1.c] = blend4(alpha, beta, p[oT + L], p[oT + R], p[oB + L], p[oB + R]);
        }
    } else {
        for (int c = 0; c < channel; ++c) dst[(int64_t)c * s1.c] = 0.0f;
    }
}

// tile of interp_backward_lds (below)
#define IB_TW 64
#define IB_TH 8
#define IB_THREADS (IB_TW * IB_TH)
#define IB_CH 3                                     // channels summed per pass
#define IB_CELLS 6144                               // 64-bit cells of LDS (49,152 bytes)

__global__ __launch_bounds__(VFI_TX * VFI_TY) void interp_backward(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ gout,
    unsigned long long* __restrict__ acc, const int* __restrict__ hdr, float* g1, float* g2, int channel, int h, int w, vfi_strides s1,
    vfi_strides s2, const int* __restrict__ tileflag) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    // (after interp_backward_lds: only the tiles that kernel left alone; VFI_TX x VFI_TY blocks nest in them)
    if (tileflag && !tileflag[(b * ((h + IB_TH - 1) / IB_TH) + y / IB_TH) * gridDim.x + blockIdx.x]) return;
    const GradAccCtx gctx = gradacc_ctx(hdr);
    const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    const float fx = flow[0];
    const float fy = flow[s2.c];
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    if (!(x2 >= 0.0f && y2 >= 0.0f && x2 < (float)w && y2 < (float)h)) return;
    const int L = (int)x2, T = (int)y2;
    const int R = min(L + 1, w - 1), Bm = min(T + 1, h - 1);
    const float alpha = x2 - (float)L, beta = y2 - (float)T;
    const float* img = in1 + (int64_t)b * s1.b;
    unsigned long long* gimg = acc + (int64_t)b * channel * h * w;       // dense [b][c][y][x] fixed-point sums (vfi_common.h)
    const float* gpx = gout + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    const int64_t oT = (int64_t)T * s1.h, oB = (int64_t)Bm * s1.h;
    const int64_t aT = (int64_t)T * w, aB = (int64_t)Bm * w;
    const float gam_y = (float)Bm - y2;         // (:161)
    const float gam_x = (float)R - x2;          // (:181)
    float botx = 0.0f, boty = 0.0f;
    for (int c = 0; c < channel; ++c) {
        const float* p = img + (int64_t)c * s1.c;
        unsigned long long* gp = gimg + (int64_t)c * h * w;
        const float g = gpx[(int64_t)c * s1.c];
        float* gfp = g1 + (int64_t)b * s1.b + (int64_t)c * s1.c;    // (the fp32 scatter of a call with non-finite inputs)
        gradacc_add(gp, gfp, aT + L, oT + L, g * (1.0f - alpha) * (1.0f - beta), gctx);     // (:151-158)
        gradacc_add(gp, gfp, aT + R, oT + R, g * alpha * (1.0f - beta), gctx);
        gradacc_add(gp, gfp, aB + L, oB + L, g * (1.0f - alpha) * beta, gctx);
        gradacc_add(gp, gfp, aB + R, oB + R, g * alpha * beta, gctx);
        const float tl = p[oT + L], tr = p[oT + R], bl = p[oB + L], br = p[oB + R];
        float temp = gam_y * (tr - tl);
        temp = fmaf(1.0f - gam_y, br - bl, temp);
        botx = fmaf(g, temp, botx);
        temp = gam_x * (bl - tl);
        temp = fmaf(1.0f - gam_x, br - tr, temp);
        boty = fmaf(g, temp, boty);
    }
    float* gf = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    gf[0] = botx;
    gf[s2.c] = boty;
}

// The image gradient through LDS, as fi_backward_ori4_lds (filterinterp.hip) does it: the addends are exact integers, so a
// 64x8 tile sums its 4 addends per pixel and channel in a window of 64-bit LDS cells over the bounding box of its taps,
// three channels at a time, and issues one global atomic per non-zero cell.  Same integer sums as the per-tap scatter,
// hence the same bits; flow gradient as in interp_backward.  A call with non-finite inputs and a tile whose window does
// not fit are flagged and left to interp_backward, launched after this kernel.
__global__ __launch_bounds__(IB_THREADS) void interp_backward_lds(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ gout,
    unsigned long long* __restrict__ acc, const int* __restrict__ hdr, int* __restrict__ tileflag, float* g2,
    int channel, int h, int w, vfi_strides s1, vfi_strides s2) {
    __shared__ unsigned long long cells[IB_CELLS];
    __shared__ int box[4];
    const int tid = threadIdx.x;
    const int x = blockIdx.x * IB_TW + (tid & (IB_TW - 1));
    const int y = blockIdx.y * IB_TH + (tid >> 6);
    const int b = blockIdx.z;
    const GradAccCtx gctx = gradacc_ctx(hdr);
    const bool inimg = x < w && y < h;
    float fx = 0.0f, fy = 0.0f;
    if (inimg) {
        const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        fx = flow[0];
        fy = flow[s2.c];
    }
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    const bool valid = inimg && x2 >= 0.0f && y2 >= 0.0f && x2 < (float)w && y2 < (float)h;
    const int L = valid ? (int)x2 : 0, T = valid ? (int)y2 : 0;
    const int R = min(L + 1, w - 1), Bm = min(T + 1, h - 1);
    if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
    __syncthreads();
    {
        const int wx0 = wave_min_i32(valid ? L : INT_MAX), wy0 = wave_min_i32(valid ? T : INT_MAX);
        const int wx1 = wave_max_i32(valid ? R : INT_MIN), wy1 = wave_max_i32(valid ? Bm : INT_MIN);
        if ((tid & 63) == 0 && wx0 != INT_MAX) {
            atomicMin(&box[0], wx0); atomicMin(&box[1], wy0);
            atomicMax(&box[2], wx1); atomicMax(&box[3], wy1);
        }
    }
    __syncthreads();
    if (box[0] == INT_MAX) return;                          // (workgroup-uniform: nothing in this tile has a gradient)
    const int bx0 = box[0], by0 = box[1], bw = box[2] - box[0] + 1, bh = box[3] - box[1] + 1;
    const int n = bw * bh;
    const int pc = min(IB_CH, IB_CELLS / n);                // channels per pass: as many as the cells allow
    if (gctx.nonfinite || pc == 0) {                        // (workgroup-uniform) left to interp_backward
        if (tid == 0) tileflag[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = 1;
        return;
    }
    const float alpha = x2 - (float)L, beta = y2 - (float)T;
    const float* img = in1 + (int64_t)b * s1.b;
    unsigned long long* gimg = acc + (int64_t)b * channel * h * w;
    const float* gpx = gout + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    const int64_t oT = (int64_t)T * s1.h, oB = (int64_t)Bm * s1.h;
    const int cT = (T - by0) * bw - bx0, cB = (Bm - by0) * bw - bx0;       // window rows' first cells, less the box's first column
    const float gam_y = (float)Bm - y2;         // (:161)
    const float gam_x = (float)R - x2;          // (:181)
    float botx = 0.0f, boty = 0.0f;
    for (int c0 = 0; c0 < channel; c0 += pc) {
        const int cn = min(pc, channel - c0);
        for (int e = tid; e < n * cn; e += IB_THREADS) cells[e] = 0ull;
        __syncthreads();
        for (int cc = 0; cc < cn && valid; ++cc) {
            const int c = c0 + cc;
            const float* p = img + (int64_t)c * s1.c;
            const float g = gpx[(int64_t)c * s1.c];
            unsigned long long* win = cells + cc * n;
            atomicAdd(&win[cT + L], (unsigned long long)__float2ll_rn(g * (1.0f - alpha) * (1.0f - beta) * gctx.scale));    // (:151-158)
            atomicAdd(&win[cT + R], (unsigned long long)__float2ll_rn(g * alpha * (1.0f - beta) * gctx.scale));
            atomicAdd(&win[cB + L], (unsigned long long)__float2ll_rn(g * (1.0f - alpha) * beta * gctx.scale));
            atomicAdd(&win[cB + R], (unsigned long long)__float2ll_rn(g * alpha * beta * gctx.scale));
            const float tl = p[oT + L], tr = p[oT + R], bl = p[oB + L], br = p[oB + R];
            float temp = gam_y * (tr - tl);
            temp = fmaf(1.0f - gam_y, br - bl, temp);
            botx = fmaf(g, temp, botx);
            temp = gam_x * (bl - tl);
            temp = fmaf(1.0f - gam_x, br - tr, temp);
            boty = fmaf(g, temp, boty);
        }
        __syncthreads();
        for (int e = tid; e < n * cn; e += IB_THREADS) {
            const unsigned long long v = cells[e];
            if (v != 0ull) {
                const int cc = e / n, r = e - cc * n;
                const int cy = r / bw, cx = r - cy * bw;
                atomicAdd(&gimg[(int64_t)(c0 + cc) * h * w + (int64_t)(by0 + cy) * w + bx0 + cx], v);
            }
        }
        __syncthreads();                                    // (the next pass zeroes the cells)
    }
    if (valid) {
        float* gf = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        gf[0] = botx;
        gf[s2.c] = boty;
    }
}

// ------------------------------------------------------------------ SeparableConv

// out[c] = sum_fy sum_fx (img[c, y+fy, x+fx] * v[fy]) * h[fx]  (:65-77).  Channels are
// processed CH at a time so v/h are fetched once per tap for all of them; each
// channel's own accumulation order is the reference's (fy outer, fx inner).
template <int CH>
__global__ __launch_bounds__(VFI_TX * VFI_TY) void sepconv_forward(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    float* __restrict__ out, int channel, int oh, int ow, int fs,
    vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides so) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= ow || y >= oh) return;
    const int b = blockIdx.z;
    const float* vp = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    const float* hp = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    for (int c0 = 0; c0 < channel; c0 += CH) {
        float acc[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[k] = 0.0f;
        const float* p = in1 + (int64_t)b * s1.b + (int64_t)c0 * s1.c + (int64_t)y * s1.h + x;
        for (int fy = 0; fy < fs; ++fy) {
            const float t2 = vp[(int64_t)fy * s2.c];
            const float* row = p + (int64_t)fy * s1.h;
            int fx = 0;
            // eight taps at a time: their 8 + 8 * CH loads are issued before the first multiply (the plain loop
            // ran one memory round trip per tap: 1.65 TMAC/s at fs = 51); the order of the sums is unchanged
            for (; fx + 8 <= fs; fx += 8) {
                float t3[8], pv[8][CH];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    t3[q] = hp[(int64_t)(fx + q) * s3.c];
#pragma unroll
                    for (int k = 0; k < CH; ++k) pv[q][k] = (c0 + k < channel) ? row[(int64_t)k * s1.c + fx + q] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int k = 0; k < CH; ++k)
                        if (c0 + k < channel) acc[k] = fmaf(pv[q][k] * t2, t3[q], acc[k]);
            }
            for (; fx < fs; ++fx) {
                const float t3 = hp[(int64_t)fx * s3.c];
#pragma unroll
                for (int k = 0; k < CH; ++k)
                    if (c0 + k < channel) acc[k] = fmaf(row[(int64_t)k * s1.c + fx] * t2, t3, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (c0 + k < channel)
                out[(int64_t)b * so.b + (int64_t)(c0 + k) * so.c + (int64_t)y * so.h + x] = acc[k];
    }
}

// Image gradient of SeparableConv (separableconv_cuda_kernel.cu:122-127 scatters it with atomics: order dependent).
// Owner computes: image cell (Y, X) receives gout[y, x] * v[fy, y, x] * h[fx, y, x] from the outputs (y, x) =
// (Y - fy, X - fx); they are summed in the raster order of (y, x) -- the order a sequential run of the reference's
// loop adds them in -- so the result is deterministic and equals the sequential sum bit for bit.
__global__ __launch_bounds__(VFI_TX * VFI_TY) void sepconv_backward_image(
    const float* __restrict__ in2, const float* __restrict__ in3, const float* __restrict__ gout, float* g1,
    int channel, int h, int w, int oh, int ow, int fs, vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides so) {
    const int X = blockIdx.x * VFI_TX + threadIdx.x;
    const int Y = blockIdx.y * VFI_TY + threadIdx.y;
    if (X >= w || Y >= h) return;
    const int b = blockIdx.z;
    for (int c = 0; c < channel; ++c) {
        float* cell = g1 + (int64_t)b * s1.b + (int64_t)c * s1.c + (int64_t)Y * s1.h + X;
        float acc = *cell;                                  // caller zero-fills; accumulate as the reference does
        for (int fy = fs - 1; fy >= 0; --fy) {              // y = Y - fy ascending
            const int y = Y - fy;
            if (y < 0 || y >= oh) continue;
            for (int fx = fs - 1; fx >= 0; --fx) {          // x = X - fx ascending
                const int x = X - fx;
                if (x < 0 || x >= ow) continue;
                const float g = gout[(int64_t)b * so.b + (int64_t)c * so.c + (int64_t)y * so.h + x];
                const float t2 = in2[(int64_t)b * s2.b + (int64_t)fy * s2.c + (int64_t)y * s2.h + x];
                const float t3 = in3[(int64_t)b * s3.b + (int64_t)fx * s3.c + (int64_t)y * s3.h + x];
                acc += g * t2 * t3;
            }
        }
        *cell = acc;
    }
}

__global__ __launch_bounds__(VFI_TX * VFI_TY) void sepconv_backward(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    const float* __restrict__ gout, float* g2, float* g3, int channel, int oh, int ow, int fs,
    vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides so) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= ow || y >= oh) return;
    const int b = blockIdx.z;
    const float* vp = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    const float* hp = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    float* gvp = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    float* ghp = g3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    for (int c = 0; c < channel; ++c) {
        const float* p = in1 + (int64_t)b * s1.b + (int64_t)c * s1.c + (int64_t)y * s1.h + x;
        const float g = gout[(int64_t)b * so.b + (int64_t)c * so.c + (int64_t)y * so.h + x];
        for (int fy = 0; fy < fs; ++fy) {
            const float t2 = vp[(int64_t)fy * s2.c];
            for (int fx = 0; fx < fs; ++fx) {               // (:114-127)
                const float t3 = hp[(int64_t)fx * s3.c];
                const float t1 = p[(int64_t)fy * s1.h + fx];
                // (the image gradient is gathered by sepconv_backward_image, in the reference's own summation order)
                // the v / h gradient cells at (y, x) belong to this thread alone
                gvp[(int64_t)fy * s2.c] += g * t1 * t3;
                ghp[(int64_t)fx * s3.c] += g * t1 * t2;
            }
        }
    }
}

// ------------------------------------------------------------------ SeparableConvFlow

__device__ __forceinline__ float com_flow(const float* __restrict__ k, int64_t cs, int fs, float* sum_out, float* mom_out) {
    float mom = 0.0f, sum = 0.0f;
    for (int f = 0; f < fs; ++f) {
        const float t = k[(int64_t)f * cs];
        mom = fmaf((float)f, t, mom);
        sum += t;
    }
    *sum_out = sum;
    *mom_out = mom;
    // the reference subtracts ((float)fs - 1.0)/2.0 in double (:75)
    return (float)((double)(mom / sum) - ((double)(float)fs - 1.0) / 2.0);
}

__global__ __launch_bounds__(VFI_TX * VFI_TY) void sepconvflow_forward(
    const float* __restrict__ in2, const float* __restrict__ in3, float* __restrict__ flow_out,
    int oh, int ow, int fs, vfi_strides s2, vfi_strides s3, vfi_strides so) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= ow || y >= oh) return;
    const int b = blockIdx.z;
    float* o = flow_out + (int64_t)b * so.b + (int64_t)y * so.h + x;
    float sum, mom;
    const float fy = com_flow(in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x, s2.c, fs, &sum, &mom);
    o[so.c] = (fabsf(sum) > 0.0f) ? fy : -2000.0f;
    const float fx = com_flow(in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x, s3.c, fs, &sum, &mom);
    o[0] = (fabsf(sum) > 0.0f) ? fx : -2000.0f;
}

__global__ __launch_bounds__(VFI_TX * VFI_TY) void sepconvflow_backward(
    const float* __restrict__ in2, const float* __restrict__ in3, const float* __restrict__ gflow,
    float* g2, float* g3, int oh, int ow, int fs, vfi_strides s2, vfi_strides s3, vfi_strides so) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= ow || y >= oh) return;
    const int b = blockIdx.z;
    const float* gpx = gflow + (int64_t)b * so.b + (int64_t)y * so.h + x;
    float sum, mom;
    com_flow(in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x, s2.c, fs, &sum, &mom);
    if (fabsf(sum) > 0.0f) {                    // plain store (:144)
        const float g = gpx[so.c];
        const float offset = mom / (sum * sum);
        float* gv = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        for (int f = 0; f < fs; ++f) gv[(int64_t)f * s2.c] = g * ((float)f / sum - offset);
    }
    com_flow(in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x, s3.c, fs, &sum, &mom);
    if (fabsf(sum) > 0.0f) {                    // accumulate (:166)
        const float g = gpx[0];
        const float offset = mom / (sum * sum);
        float* gh = g3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
        for (int f = 0; f < fs; ++f) gh[(int64_t)f * s3.c] = fmaf(g, (float)f / sum - offset, gh[(int64_t)f * s3.c]);
    }
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_interpolation_forward(const float* input1, const float* input2, float* output, int batch,
                                          int channel, int h, int w, vfi_strides s1, vfi_strides s2,
                                          vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !output) return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(interp_forward, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input1, input2, output, channel, h, w, s1, s2);
    return launch_status();
}

extern "C" int vfi_interpolation_backward(const float* input1, const float* input2, const float* gradoutput,
                                           float* gradinput1, float* gradinput2, int batch, int channel, int h, int w,
                                           vfi_strides s1, vfi_strides s2, vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !gradoutput || !gradinput1 ||
        !gradinput2)
        return VFI_ERR_SHAPE;
    unsigned long long* acc;
    int* hdr;
    // (the tap weights are bilinear fractions: at most 1)
    int* flags = nullptr;                                   // one word per 64x8 tile: "the staged kernel left it alone"
    const dim3 tiles((w + IB_TW - 1) / IB_TW, (h + IB_TH - 1) / IB_TH, batch);
    const int err = gradacc_begin((hipStream_t)stream, gradoutput, batch, channel, h, w, s1, nullptr, 0, s1, &acc, &hdr,
                                  (int)(tiles.x * tiles.y * tiles.z), &flags);
    if (err != VFI_OK) return err;
    static_assert(VFI_TX == IB_TW && IB_TH % VFI_TY == 0, "interp_backward's blocks nest in the staged kernel's tiles");
    hipLaunchKernelGGL(interp_backward_lds, tiles, dim3(IB_THREADS), 0, (hipStream_t)stream,
                       input1, input2, gradoutput, acc, hdr, flags, gradinput2, channel, h, w, s1, s2);
    hipLaunchKernelGGL(interp_backward, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input1, input2, gradoutput, acc, hdr, gradinput1, gradinput2, channel, h, w, s1, s2, flags);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    return gradacc_finish((hipStream_t)stream, acc, hdr, gradinput1, batch, channel, h, w, s1);
}

extern "C" int vfi_separableconv_forward(const float* input1, const float* input2, const float* input3, float* output,
                                          int batch, int channel, int h, int w, int filter_size, vfi_strides s1,
                                          vfi_strides s2, vfi_strides s3, vfi_strides so, vfi_stream_t stream) {
    const int oh = h - filter_size + 1, ow = w - filter_size + 1;
    if (batch <= 0 || channel <= 0 || filter_size <= 0 || oh <= 0 || ow <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !output) return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(sepconv_forward<3>, pixel_grid(ow, oh, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input1, input2, input3, output, channel, oh, ow, filter_size, s1, s2, s3, so);
    return launch_status();
}

extern "C" int vfi_separableconv_backward(const float* input1, const float* input2, const float* input3,
                                           const float* gradoutput, float* gradinput1, float* gradinput2,
                                           float* gradinput3, int batch, int channel, int h, int w, int filter_size,
                                           vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides so,
                                           vfi_stream_t stream) {
    const int oh = h - filter_size + 1, ow = w - filter_size + 1;
    if (batch <= 0 || channel <= 0 || filter_size <= 0 || oh <= 0 || ow <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !gradoutput || !gradinput1 || !gradinput2 || !gradinput3)
        return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(sepconv_backward, pixel_grid(ow, oh, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input1, input2, input3, gradoutput, gradinput2, gradinput3, channel, oh, ow,
                       filter_size, s1, s2, s3, so);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    hipLaunchKernelGGL(sepconv_backward_image, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input2, input3, gradoutput, gradinput1, channel, h, w, oh, ow, filter_size, s1, s2, s3, so);
    return launch_status();
}

extern "C" int vfi_separableconvflow_forward(const float* input2, const float* input3, float* flow_output, int batch,
                                              int h, int w, int filter_size, vfi_strides s2, vfi_strides s3,
                                              vfi_strides so, vfi_stream_t stream) {
    const int oh = h - filter_size + 1, ow = w - filter_size + 1;
    if (batch <= 0 || filter_size <= 0 || oh <= 0 || ow <= 0 || !input2 || !input3 || !flow_output)
        return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(sepconvflow_forward, pixel_grid(ow, oh, batch), dim3(VFI_TX, VFI_TY, 1), 0,
                       (hipStream_t)stream, input2, input3, flow_output, oh, ow, filter_size, s2, s3, so);
    return launch_status();
}

extern "C" int vfi_separableconvflow_backward(const float* input2, const float* input3, const float* gradflow_output,
                                               float* gradinput2, float* gradinput3, int batch, int h, int w,
                                               int filter_size, vfi_strides s2, vfi_strides s3, vfi_strides so,
                                               vfi_stream_t stream) {
    const int oh = h - filter_size + 1, ow = w - filter_size + 1;
    if (batch <= 0 || filter_size <= 0 || oh <= 0 || ow <= 0) return VFI_ERR_SHAPE;
    if (!input2 || !input3 || !gradflow_output || !gradinput2 || !gradinput3) return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(sepconvflow_backward, pixel_grid(ow, oh, batch), dim3(VFI_TX, VFI_TY, 1), 0,
                       (hipStream_t)stream, input2, input3, gradflow_output, gradinput2, gradinput3, oh, ow,
                       filter_size, s2, s3, so);
    return launch_status();
}
