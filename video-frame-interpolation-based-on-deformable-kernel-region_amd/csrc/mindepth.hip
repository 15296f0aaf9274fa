// mindepth.hip -- MinDepthFlowProjection for gfx950.
//
// Replaces mindepthflowprojection_cuda_kernel.cu:27-331 of the reference.  There every source
// pixel whose projected position lands inside the frame writes (-fx,-fy) and its weight
// (input2 = inverse depth) to the top-left integer target if the weight beats the one recorded
// in `count` -- as an unguarded read / compare / write (:76-84), so concurrent sources of one
// target race and the reference's result depends on scheduling.  This library defines the
// result as that of the same statements executed sequentially in raster order of the sources
// (the reference's CPU-thinkable order): the target keeps the source with the LARGEST weight
// that exceeds the incoming `count`, the FIRST such source in raster order on ties.  It is
// computed without a race: one 64-bit atomicMax per source on a key
//     (order-preserving bits of the weight) << 32 | (0xffffffff - source index)
// in a per-stream scratch plane, then one pass that decodes the winner per target and leaves bitmaps of
// "count != 0" for the hole filler (a walk to the nearest non-hole is then a few word loads).  Hole
// filling (:121-206) and backward (:209-331) follow the reference statement by statement; the
// backward compares the weight with `count` at all four neighbours although the forward writes
// only the top-left one (:271-286), and leaves gradinput2 untouched (:289-326 are comments).

#include "vfi_common.h"
#include "bitwalk.h"
#include "workspace.h"

#include <limits.h>

namespace vfi {

// monotone map float -> uint32 (total order of the values, -0 < +0)
__device__ __forceinline__ uint32_t md_order_bits(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float md_order_value(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(VFI_TX * VFI_TY) void mindepth_bid(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ count,
    unsigned long long* __restrict__ keys, int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides sc) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* flow = in1 + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    const float fx = flow[0];
    const float fy = flow[s1.c];
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(w - 1) && y2 <= (float)(h - 1))) return;   // (:68)
    const int L = (int)x2, T = (int)y2;
    const float weight = in2[(int64_t)b * s2.b + (int64_t)y * s2.h + x];
    if (!(weight > count[(int64_t)b * sc.b + (int64_t)T * sc.h + L])) return;                    // (:78-79)
    const unsigned long long key =
        ((unsigned long long)md_order_bits(weight) << 32) | (0xffffffffu - (uint32_t)(y * w + x));
    atomicMax(keys + ((int64_t)b * h + T) * w + L, key);
}

// bitmaps of "count != 0" for the hole filler: rowmap[b][y][x / 32] and colmap[b][x][y / 32]
struct MdMaps { int rmw, cmw, rowmap, colmap; };

// Decodes the winner of every target and leaves the two bitmaps.  A workgroup covers 64 x 32 targets (a wave:
// eight rows of 64), so that a column's 32 row bits are one word assembled in LDS and stored without atomics.
#define MD_ROWS 32
__global__ __launch_bounds__(256) void mindepth_award(
    const float* __restrict__ in1, const unsigned long long* __restrict__ keys, float* __restrict__ count,
    float* __restrict__ out, int* __restrict__ bits, MdMaps m, int h, int w, vfi_strides s1, vfi_strides sc) {
    __shared__ unsigned colm[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int b = blockIdx.z;
    if (threadIdx.x < 64) colm[threadIdx.x] = 0u;
    __syncthreads();
    unsigned mine = 0u;
#pragma unroll
    for (int r = 0; r < MD_ROWS / 4; ++r) {
        const int yl = wave * (MD_ROWS / 4) + r, y = blockIdx.y * MD_ROWS + yl;
        bool nz = false;
        if (x < w && y < h) {
            const unsigned long long key = keys[((int64_t)b * h + y) * w + x];
            float* cn = count + (int64_t)b * sc.b + (int64_t)y * sc.h + x;
            float c = *cn;                                  // nobody beat the incoming count: both stay untouched
            if (key != 0ull) {
                const uint32_t src = 0xffffffffu - (uint32_t)key;
                const int sy = (int)(src / (uint32_t)w), sx = (int)(src % (uint32_t)w);
                const float* flow = in1 + (int64_t)b * s1.b + (int64_t)sy * s1.h + sx;
                float* o = out + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
                o[0] = -flow[0];                            // (:80-82)
                o[s1.c] = -flow[s1.c];
                c = md_order_value((uint32_t)(key >> 32));
                *cn = c;
            }
            nz = c != 0.0f;
        }
        const unsigned long long rowbits = __ballot(nz);
        if (lane < 2 && y < h && (int)blockIdx.x * 2 + lane < m.rmw)
            bits[m.rowmap + (b * h + y) * m.rmw + blockIdx.x * 2 + lane] = (int)(unsigned)(rowbits >> (32 * lane));
        if (nz) mine |= 1u << yl;
    }
    if (mine) atomicOr(&colm[lane], mine);
    __syncthreads();
    if (threadIdx.x < 64 && x < w) bits[m.colmap + (b * w + x) * m.cmw + blockIdx.y] = (int)colm[threadIdx.x];
}

// (:121-206) a hole (count <= 0) takes the plain mean of the values at the nearest cells with count != 0 to its
// left / right / above / below, those with count > 0 only.  The reference walks there cell by cell (a dependent
// chain of loads as long as the uncovered strip: 0.34 ms at 1080p when this kernel did the same); the bitmaps
// make a walk a few word loads.  Holes only read non-holes and only holes are written: no ordering hazard.
__global__ __launch_bounds__(VFI_TX * VFI_TY) void mindepth_fillhole(
    const float* __restrict__ count, float* out, const int* __restrict__ bits, MdMaps m, int h, int w,
    vfi_strides s1, vfi_strides sc) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* cn = count + (int64_t)b * sc.b;
    if (!(cn[(int64_t)y * sc.h + x] <= 0.0f)) return;
    const int* rl = bits + m.rowmap + (b * h + y) * m.rmw;
    const int* cl = bits + m.colmap + (b * w + x) * m.cmw;
    const int xl = proj_bit_walk(rl, x, w, -1), xr = proj_bit_walk(rl, x, w, +1);
    const int yu = proj_bit_walk(cl, y, h, -1), yd = proj_bit_walk(cl, y, h, +1);
    // a walk that found nothing contributes weight 0; its position only has to be valid
    const int lo = xl < 0 ? x : xl, ro = xr < 0 ? x : xr, uo = yu < 0 ? y : yu, dn = yd < 0 ? y : yd;
    float lt = xl < 0 ? 0.0f : cn[(int64_t)y * sc.h + xl];
    float rt = xr < 0 ? 0.0f : cn[(int64_t)y * sc.h + xr];
    float ut = yu < 0 ? 0.0f : cn[(int64_t)yu * sc.h + x];
    float dt = yd < 0 ? 0.0f : cn[(int64_t)yd * sc.h + x];
    if (lt + rt + ut + dt <= 0.0f) return;                  // (:175-178)
    lt = lt > 0.0f ? 1.0f : 0.0f;
    rt = rt > 0.0f ? 1.0f : 0.0f;
    ut = ut > 0.0f ? 1.0f : 0.0f;
    dt = dt > 0.0f ? 1.0f : 0.0f;
    float* o = out + (int64_t)b * s1.b;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        float* p = o + (int64_t)ch * s1.c;
        float acc = lt * p[(int64_t)y * s1.h + lo];         // (:185-203) the weights are 0 / 1: products exact
        acc = fmaf(rt, p[(int64_t)y * s1.h + ro], acc);
        acc = fmaf(ut, p[(int64_t)uo * s1.h + x], acc);
        acc = fmaf(dt, p[(int64_t)dn * s1.h + x], acc);
        p[(int64_t)y * s1.h + x] = acc / (lt + rt + ut + dt);
    }
}

__global__ __launch_bounds__(VFI_TX * VFI_TY) void mindepth_backward(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ count,
    const float* __restrict__ gout, float* g1, int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides sc) {
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
    const float weight = in2[(int64_t)b * s2.b + (int64_t)y * s2.h + x];
    const int ty[4] = {T, T, Bm, Bm}, tx[4] = {L, R, L, R};
    const float* cn = count + (int64_t)b * sc.b;
    const float* go = gout + (int64_t)b * s1.b;
    float* g = g1 + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    float a0 = g[0], a1 = g[s1.c];                          // caller zero-fills; accumulate as the reference does
#pragma unroll
    for (int k = 0; k < 4; ++k)                             // (:271-286)
        if (weight == cn[(int64_t)ty[k] * sc.h + tx[k]]) {
            a0 += -go[(int64_t)ty[k] * s1.h + tx[k]];
            a1 += -go[s1.c + (int64_t)ty[k] * s1.h + tx[k]];
        }
    g[0] = a0;
    g[s1.c] = a1;
}

static unsigned long long* mindepth_keys(hipStream_t st, size_t n) {
    return static_cast<unsigned long long*>(ws_get(st, WS_MINDEPTH, n * sizeof(unsigned long long), false, nullptr));
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_mindepthflowprojection_forward(const float* input1, const float* input2, float* count, float* output,
                                                   int batch, int h, int w, int fillhole, vfi_strides s1,
                                                   vfi_strides s2, vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !count || !output) return VFI_ERR_SHAPE;
    if ((int64_t)h * w > 0xffffffffll) return VFI_ERR_SHAPE;             // the key holds a 32-bit source index
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)batch * h * w;
    MdMaps m;
    m.rmw = (w + 31) / 32;
    m.cmw = (h + MD_ROWS - 1) / MD_ROWS;
    const size_t bit_words = (size_t)batch * ((size_t)h * m.rmw + (size_t)w * m.cmw);
    if (bit_words > (size_t)INT_MAX) return VFI_ERR_SHAPE;
    m.rowmap = 0;
    m.colmap = batch * h * m.rmw;
    unsigned long long* keys = mindepth_keys(st, n + (bit_words + 1) / 2);
    if (!keys) return VFI_ERR_LAUNCH;
    int* bits = reinterpret_cast<int*>(keys + n);           // every word is written by mindepth_award
    if (hipMemsetAsync(keys, 0, n * sizeof(unsigned long long), st) != hipSuccess) return VFI_ERR_LAUNCH;
    const dim3 grid = pixel_grid(w, h, batch), block(VFI_TX, VFI_TY, 1);
    hipLaunchKernelGGL(mindepth_bid, grid, block, 0, st, input1, input2, count, keys, h, w, s1, s2, sc);
    hipLaunchKernelGGL(mindepth_award, dim3((unsigned)((w + 63) / 64), (unsigned)m.cmw, (unsigned)batch), dim3(256), 0, st,
                       input1, keys, count, output, bits, m, h, w, s1, sc);
    if (fillhole) hipLaunchKernelGGL(mindepth_fillhole, grid, block, 0, st, count, output, bits, m, h, w, s1, sc);
    return launch_status();
}

extern "C" int vfi_mindepthflowprojection_backward(const float* input1, const float* input2, const float* count,
                                                    const float* gradoutput, float* gradinput1, int batch, int h, int w,
                                                    vfi_strides s1, vfi_strides s2, vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !count || !gradoutput || !gradinput1) return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(mindepth_backward, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input1, input2, count, gradoutput, gradinput1, h, w, s1, s2, sc);
    return launch_status();
}
