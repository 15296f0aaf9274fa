// filterinterp.hip -- adaptive-warping layer (FilterInterpolation) for gfx950.
//
// Semantics: filterinterpolation_cuda_kernel.cu:2692-2823 (_ori forward),
// :2827-3125 (_ori backward), :29-426 / :1353-1496 / :2070-2191 (deformable
// forwards) of the reference; entry points replace filterinterpolation_cuda.cc.
//
// This file holds the direct-gather kernels: one thread per output pixel, a
// wave covers 64 consecutive x so flow / filter / output planes move as full
// 256-B rows; flow, the blend weights and (fs == 4) all 16 filter taps live in
// registers across the channel loop (the reference re-fetches them per channel).
// The LDS-staged forward for fs == 4 lives in filterinterp_lds.hip and falls
// back to the kernel here when a tile's tap window does not fit its LDS budget.
#include "filterinterp_dev.h"

#include <limits.h>

namespace vfi {

// ------------------------------------------------------------------ forward, _ori

template <bool FS4>
__global__ __launch_bounds__(VFI_TX * VFI_TY) void fi_forward_ori_direct(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    float* __restrict__ out, int channel, int h, int w, int fs,
    vfi_strides s1, vfi_strides s2, vfi_strides s3) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    const float fx = flow[0];
    const float fy = flow[s2.c];
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    const float* img = in1 + (int64_t)b * s1.b;
    float* dst = out + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    if (!fi_valid(fx, fy, x2, y2, w, h)) {
        // copy-through (:2814-2818)
        const float* src = img + (int64_t)y * s1.h + x;
        for (int c = 0; c < channel; ++c) dst[(int64_t)c * s1.c] = src[(int64_t)c * s1.c];
        return;
    }
    const int ix = (int)x2, iy = (int)y2;
    const int L = ix + 1 - fs / 2, T = iy + 1 - fs / 2;
    const float alpha = x2 - (float)ix;
    const float beta = y2 - (float)iy;
    const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    if constexpr (FS4) {
        // fs == 4: every quadrant is 2x2; taps, clamped rows and columns hoisted out of the channel loop
        float f[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = fpx[(int64_t)k * s3.c];
        fi4_channels_direct(img, dst, 0, channel, s1.c, (int)s1.h, h, w, L, T, f, alpha, beta);
    } else {
        for (int c = 0; c < channel; ++c) {
            float q[4];
            quadrants_generic(img + (int64_t)c * s1.c, fpx, s3.c, (int)s1.h, h, w, fs, L, T, ix, iy, q);
            dst[(int64_t)c * s1.c] = blend4(alpha, beta, q[0], q[1], q[2], q[3]);
        }
    }
}

// ------------------------------------------------------------------ backward, _ori

// tile of fi_backward_ori4_lds (below)
#define FB_TW 64
#ifndef FB_TH
#define FB_TH 8
#endif
#ifndef FB_WAVES
#define FB_WAVES 4                                  // (launch bound: waves per SIMD the kernel must fit -- two workgroups per CU)
#endif
#define FB_THREADS (FB_TW * FB_TH)
#define FB_CH 3                                     // channels summed per pass
#define FB_CELLS 6144                               // 64-bit gradient cells per pass (49,152 bytes); half as many bytes of image windows
// what the launch bound promises must also hold for the LDS array: FB_WAVES waves per SIMD = FB_WAVES * 256 / FB_THREADS
// workgroups per CU, each with (2 + 1.5 FB_CELLS) * 8 bytes of the CU's 160 KB (a build that overrides FB_TH or FB_WAVES
// would otherwise keep compiling and silently lose occupancy)
static_assert(FB_THREADS <= 1024 && FB_THREADS % 64 == 0, "workgroup size");
static_assert(((FB_WAVES * 256) / FB_THREADS) * (2 + FB_CELLS + FB_CELLS / 2) * 8 <= 160 * 1024, "workgroups per CU x LDS per workgroup exceed the CU's LDS");

__global__ __launch_bounds__(VFI_TX * VFI_TY) void fi_backward_ori(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    const float* __restrict__ gout, unsigned long long* __restrict__ acc, const int* __restrict__ hdr, float* g1, float* g2, float* g3,
    int channel, int h, int w, int fs, vfi_strides s1, vfi_strides s2, vfi_strides s3, const int* __restrict__ tileflag) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    // (after fi_backward_ori4_lds: only the tiles that kernel left alone; VFI_TX x VFI_TY blocks nest in them)
    if (tileflag && !tileflag[(b * ((h + FB_TH - 1) / FB_TH) + y / FB_TH) * gridDim.x + blockIdx.x]) return;
    const GradAccCtx gctx = gradacc_ctx(hdr);
    const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    const float fx = flow[0];
    const float fy = flow[s2.c];
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    if (!fi_valid(fx, fy, x2, y2, w, h)) return;           // no gradient (:2863-2864)
    const int ix = (int)x2, iy = (int)y2;
    const int L = ix + 1 - fs / 2, T = iy + 1 - fs / 2;
    const int R = L + fs, Bm = T + fs;
    const float alpha = x2 - (float)ix;
    const float beta = y2 - (float)iy;
    const float* img = in1 + (int64_t)b * s1.b;
    unsigned long long* gimg = acc + (int64_t)b * channel * h * w;       // dense [b][c][y][x] fixed-point sums
    const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    float* gfpx = g3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    const float* gpx = gout + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    float gx = 0.0f, gy = 0.0f;
    for (int c = 0; c < channel; ++c) {
        const float* p = img + (int64_t)c * s1.c;
        unsigned long long* gp = gimg + (int64_t)c * h * w;
        float* gfp = g1 + (int64_t)b * s1.b + (int64_t)c * s1.c;    // (the fp32 scatter of a call with non-finite inputs)
        const float g = gpx[(int64_t)c * s1.c];
        const float qg[4] = { g * (1.0f - alpha) * (1.0f - beta), g * alpha * (1.0f - beta),
                              g * (1.0f - alpha) * beta,          g * alpha * beta };
        float q[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (int quad = 0; quad < 4; ++quad) {
            const int j0 = (quad < 2) ? T : iy + 1, j1 = (quad < 2) ? iy : Bm - 1;
            const int i0 = (quad & 1) ? ix + 1 : L, i1 = (quad & 1) ? R - 1 : ix;
            float acc = 0.0f;
            for (int j = j0; j <= j1; ++j) {
                const int64_t ro = (int64_t)clampi(j, 0, h - 1) * s1.h;
                for (int i = i0; i <= i1; ++i) {
                    const int64_t o = ro + clampi(i, 0, w - 1);
                    const int64_t k = (int64_t)((j - T) * fs + (i - L)) * s3.c;
                    const float pv = p[o], fv = fpx[k];
                    // image gradient: other pixels hit the same cell -> order-free fixed-point atomic (vfi_common.h).
                    // The filter gradient cell belongs to this thread alone (index is this pixel's own), so a plain
                    // read-modify-write is equivalent to the reference's atomicAdd.
                    gradacc_add(gp, gfp, (int64_t)clampi(j, 0, h - 1) * w + clampi(i, 0, w - 1), o, qg[quad] * fv, gctx);
                    gfpx[k] += qg[quad] * pv;
                    acc = fmaf(pv, fv, acc);
                }
            }
            q[quad] = acc;
        }
        {   // flow gradient by quadrant differences (:2965-3102)
            const float gamma = 1.0f - beta;
            float temp = gamma * (q[1] - q[0]);
            temp = fmaf(1.0f - gamma, q[3] - q[2], temp);
            gx = fmaf(g, temp, gx);
        }
        {
            const float gamma = 1.0f - alpha;
            float temp = gamma * (q[2] - q[0]);
            temp = fmaf(1.0f - gamma, q[3] - q[1], temp);
            gy = fmaf(g, temp, gy);
        }
    }
    float* gf = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    gf[0] = gx;
    gf[s2.c] = gy;
}

// fs == 4, image values and image gradient through LDS (round 3).  The per-tap kernel above gathers 16 image values per
// pixel and channel from global memory and scatters 16 global 64-bit atomics (110 M of them at 1080p, C = 3).  Here a
// workgroup owns a 64x8 tile and works through the channels three at a time.  The taps of `_ori` are undisplaced, so the
// bounding box of the tile's (clamped) taps is both the window of image values it reads and the window of gradient cells
// it writes: the three planes' windows are staged by LDS-DMA, the addends -- exact integers, summable in any grouping --
// are added to 64-bit cells in LDS, and each non-zero cell leaves with ONE global atomic ((64 + 12) x (8 + 12) cells
// instead of 512 x 16 addends per tile and channel).  Taps outside, the pass's channels inside; the 16 filter taps and
// the 16 filter-gradient sums of the pixel live in registers (the per-tap kernel re-reads the taps and read-modify-writes
// the gradient cells in global memory once per channel and tap).  Every sum keeps the per-tap kernel's order and starting
// value: its bits.  A call with non-finite inputs (fp32 atomics: vfi_common.h) and a tile whose window does not fit are
// flagged, and fi_backward_ori, launched after this kernel, does those tiles only.  A tile with a large window takes two
// channels, or one, per pass.
typedef __attribute__((address_space(3))) void* fb_lptr_t;

__global__ __launch_bounds__(FB_THREADS, FB_WAVES) void fi_backward_ori4_lds(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    const float* __restrict__ gout, unsigned long long* __restrict__ acc, const int* __restrict__ hdr, int* __restrict__ tileflag,
    float* g2, float* g3, int channel, int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides s3) {
    constexpr int fs = 4;
    // one array: header (bounding box), gradient cells, image windows -- FB_CELLS of each for a pass
    __shared__ __attribute__((aligned(16))) unsigned long long lds64[2 + FB_CELLS + FB_CELLS / 2];
    int* box = reinterpret_cast<int*>(lds64);
    unsigned long long* cells = lds64 + 2;
    float* wins = reinterpret_cast<float*>(lds64 + 2 + FB_CELLS);
    const int tid = threadIdx.x;
    const int x = blockIdx.x * FB_TW + (tid & (FB_TW - 1));
    const int y = blockIdx.y * FB_TH + (tid >> 6);
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
    const bool valid = inimg && fi_valid(fx, fy, x2, y2, w, h);     // an invalid pixel has no gradient (:2863-2864)
    const int ix = valid ? (int)x2 : 0, iy = valid ? (int)y2 : 0;
    const int L = ix - 1, T = iy - 1;
    const float alpha = x2 - (float)ix;
    const float beta = y2 - (float)iy;
    int ro[4], co[4];                                       // clamped rows and columns of the pixel's window
#pragma unroll
    for (int k = 0; k < 4; ++k) { ro[k] = clampi(T + k, 0, h - 1); co[k] = clampi(L + k, 0, w - 1); }
    // ---- bounding box of the tile's (clamped) taps
    if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
    __syncthreads();
    {
        const int wx0 = wave_min_i32(valid ? co[0] : INT_MAX), wy0 = wave_min_i32(valid ? ro[0] : INT_MAX);
        const int wx1 = wave_max_i32(valid ? co[3] : INT_MIN), wy1 = wave_max_i32(valid ? ro[3] : INT_MIN);
        if ((tid & 63) == 0 && wx0 != INT_MAX) {
            atomicMin(&box[0], wx0); atomicMin(&box[1], wy0);
            atomicMax(&box[2], wx1); atomicMax(&box[3], wy1);
        }
    }
    __syncthreads();
    if (box[0] == INT_MAX) return;                          // (workgroup-uniform: nothing in this tile has a gradient)
    const int bx0 = box[0], by0 = box[1], bw = box[2] - box[0] + 1, bh = box[3] - box[1] + 1;
    const int n = bw * bh;
    const int hs = (int)s1.h;
    // channels per pass: as many (at most FB_CH) as fit the cells and the window slots -- a slot is the window rounded up to
    // whole DMA instructions, so that a window's last instruction stays inside its slot
    const int slot_floats = (n + FB_THREADS - 1) & ~(FB_THREADS - 1);
    const int pc = min(FB_CH, FB_CELLS / slot_floats);
    if (gctx.nonfinite || pc == 0 || (int64_t)h * hs * 4 > INT_MAX) {     // (workgroup-uniform) left to fi_backward_ori
        if (tid == 0) tileflag[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = 1;
        return;
    }

    const float* img = in1 + (int64_t)b * s1.b;
    unsigned long long* gimg = acc + (int64_t)b * channel * h * w;       // dense [b][c][y][x] fixed-point sums
    const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    float* gfpx = g3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    const float* gpx = gout + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    float fv[16], gf16[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { fv[k] = valid ? fpx[(int64_t)k * s3.c] : 0.0f; gf16[k] = valid ? gfpx[(int64_t)k * s3.c] : 0.0f; }
    int lrow[4];                                            // the window rows' first cells, less the box's first column
#pragma unroll
    for (int k = 0; k < 4; ++k) lrow[k] = (ro[k] - by0) * bw - bx0;
    const int plane_bytes = 4 * ((h - 1) * hs + w);
    const float inv_bw = 1.0f / (float)bw;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
    float gx = 0.0f, gy = 0.0f;
    for (int c0 = 0; c0 < channel; c0 += pc) {
        const int cn = min(pc, channel - c0);
        // ---- stage the pass's windows (element e of a window = cell e: row-major, bw per row, all inside the image), zero the
        // cells, fetch gradoutput
        for (int cc = 0; cc < cn; ++cc) {
            const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)(c0 + cc) * s1.c), 0, plane_bytes, 0x00020000);
            float* slot = wins + cc * slot_floats + wave_first;
            for (int e0 = 0; e0 < n; e0 += FB_THREADS) {
                const int e = e0 + tid;
                const int r = fi_row_of(e, inv_bw);
                const int col = e - r * bw;
                // (lanes past the window get an out-of-range offset: they write zeros, inside the window's own slot)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (fb_lptr_t)(slot + e0), 4, e < n ? 4u * (unsigned)((by0 + r) * hs + bx0 + col) : 0x80000000u, 0, 0, 0);
            }
        }
        for (int e = tid; e < n * cn; e += FB_THREADS) cells[e] = 0ull;
        float gv[FB_CH];
#pragma unroll
        for (int cc = 0; cc < FB_CH; ++cc) gv[cc] = (valid && cc < cn) ? gpx[(int64_t)(c0 + cc) * s1.c] : 0.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (valid) {
            float qg[FB_CH][4], q[FB_CH][4];
#pragma unroll
            for (int cc = 0; cc < FB_CH; ++cc) {
                const float g = gv[cc];
                qg[cc][0] = g * (1.0f - alpha) * (1.0f - beta); qg[cc][1] = g * alpha * (1.0f - beta);
                qg[cc][2] = g * (1.0f - alpha) * beta;          qg[cc][3] = g * alpha * beta;
                q[cc][0] = q[cc][1] = q[cc][2] = q[cc][3] = 0.0f;
            }
            // taps in the per-tap kernel's order (quadrant by quadrant), the pass's channels inside
#pragma unroll
            for (int quad = 0; quad < 4; ++quad)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const int j = (quad >> 1) * 2 + jj, i = (quad & 1) * 2 + ii, k = j * fs + i;
                        const int cell = lrow[j] + co[i];
#pragma unroll
                        for (int cc = 0; cc < FB_CH; ++cc) {
                            if (cc < cn) {                  // (workgroup-uniform)
                                const float pv = wins[cc * slot_floats + cell];
                                atomicAdd(&cells[cc * n + cell], (unsigned long long)__float2ll_rn(qg[cc][quad] * fv[k] * gctx.scale));
                                gf16[k] += qg[cc][quad] * pv;
                                q[cc][quad] = fmaf(pv, fv[k], q[cc][quad]);
                            }
                        }
                    }
#pragma unroll
            for (int cc = 0; cc < FB_CH; ++cc) {
                if (cc < cn) {
                    const float g = gv[cc];
                    {   // flow gradient by quadrant differences (:2965-3102)
                        const float gamma = 1.0f - beta;
                        float temp = gamma * (q[cc][1] - q[cc][0]);
                        temp = fmaf(1.0f - gamma, q[cc][3] - q[cc][2], temp);
                        gx = fmaf(g, temp, gx);
                    }
                    {
                        const float gamma = 1.0f - alpha;
                        float temp = gamma * (q[cc][2] - q[cc][0]);
                        temp = fmaf(1.0f - gamma, q[cc][3] - q[cc][1], temp);
                        gy = fmaf(g, temp, gy);
                    }
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < n * cn; e += FB_THREADS) {
            const unsigned long long v = cells[e];
            if (v != 0ull) {
                const int cc = e / n, r = e - cc * n;
                const int cy = r / bw, cx = r - cy * bw;
                atomicAdd(&gimg[(int64_t)(c0 + cc) * h * w + (int64_t)(by0 + cy) * w + bx0 + cx], v);
            }
        }
        __syncthreads();                                    // (the next pass overwrites windows and cells)
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < 16; ++k) gfpx[(int64_t)k * s3.c] = gf16[k];
        float* gf = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        gf[0] = gx;
        gf[s2.c] = gy;
    }
}

// ------------------------------------------------------------------ forward, deformable variants

// bilinear sample at (clamped tap + learned offset) (:98-111); the four corner
// indices are clamped to the image (the reference leaves them unclamped: UB there)
__device__ __forceinline__ float defor_tap(const float* __restrict__ p, int hs, int h, int w, float fracY, float fracX) {
    const int Top = (int)fracY, Left = (int)fracX;
    const float phiY = fracY - (float)Top;
    const float phiX = fracX - (float)Left;
    const float PTL = (1.0f - phiX) * (1.0f - phiY);
    const float PTR = phiX * (1.0f - phiY);
    const float PBL = (1.0f - phiX) * phiY;
    const float PBR = phiY * phiX;
    const int64_t t = (int64_t)clampi(Top, 0, h - 1) * hs, bo = (int64_t)clampi(Top + 1, 0, h - 1) * hs;
    const int l = clampi(Left, 0, w - 1), r = clampi(Left + 1, 0, w - 1);
    float s = PTL * p[t + l];
    s = fmaf(PTR, p[t + r], s);
    s = fmaf(PBL, p[bo + l], s);
    s = fmaf(PBR, p[bo + r], s);
    return s;
}

template <int VARIANT>
__global__ __launch_bounds__(VFI_TX * VFI_TY) void fi_forward_defor(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    const float* __restrict__ in4, float* __restrict__ out, int channel, int h, int w, int fs,
    vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
    const float fx = flow[0];
    const float fy = flow[s2.c];
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    const float* img = in1 + (int64_t)b * s1.b;
    float* dst = out + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    if (!fi_valid(fx, fy, x2, y2, w, h)) {
        const float* src = img + (int64_t)y * s1.h + x;
        for (int c = 0; c < channel; ++c) dst[(int64_t)c * s1.c] = src[(int64_t)c * s1.c];
        return;
    }
    const int fs2 = fs * fs;
    const int ix = (int)x2, iy = (int)y2;
    const int L = ix + 1 - fs / 2, T = iy + 1 - fs / 2;
    const int R = L + fs, Bm = T + fs;
    const float alpha = x2 - (float)ix;
    const float beta = y2 - (float)iy;
    // VARIANT 2: the third input IS the offset field, there are no filter weights
    const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    const float* opx = (VARIANT == VFI_DEFOR_NOFILTER) ? fpx : in4 + (int64_t)b * s4.b + (int64_t)y * s4.h + x;
    const int64_t ocs = (VARIANT == VFI_DEFOR_NOFILTER) ? s3.c : s4.c;
    for (int c = 0; c < channel; ++c) {
        const float* p = img + (int64_t)c * s1.c;
        float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
        if constexpr (VARIANT == VFI_DEFOR_OFFSET) {
            for (int quad = 0; quad < 4; ++quad) {
                const int j0 = (quad < 2) ? T : iy + 1, j1 = (quad < 2) ? iy : Bm - 1;
                const int i0 = (quad & 1) ? ix + 1 : L, i1 = (quad & 1) ? R - 1 : ix;
                float acc = 0.0f;
                for (int j = j0; j <= j1; ++j) {
                    const int cj = clampi(j, 0, h - 1);
                    for (int i = i0; i <= i1; ++i) {
                        const int ci = clampi(i, 0, w - 1);
                        const int k = (j - T) * fs + (i - L);
                        const float fracY = (float)cj + opx[(int64_t)k * ocs];
                        const float fracX = (float)ci + opx[(int64_t)(fs2 + k) * ocs];
                        acc = fmaf(defor_tap(p, (int)s1.h, h, w, fracY, fracX), fpx[(int64_t)k * s3.c], acc);
                    }
                }
                if (quad == 0) TL = acc; else if (quad == 1) TR = acc; else if (quad == 2) BL = acc; else BR = acc;
            }
        } else {
            for (int j = T; j < Bm; ++j) {
                const int cj = clampi(j, 0, h - 1);
                for (int i = L; i < R; ++i) {
                    const int ci = clampi(i, 0, w - 1);
                    const int k = (j - T) * fs + (i - L);
                    const float fracY = (float)cj + opx[(int64_t)k * ocs];
                    const float fracX = (float)ci + opx[(int64_t)(fs2 + k) * ocs];
                    const float v = defor_tap(p, (int)s1.h, h, w, fracY, fracX);
                    if constexpr (VARIANT == VFI_DEFOR_NOFILTER) {
                        if (fracX <= x2 && fracY <= y2) TL = TL + v;
                        if (fracX >  x2 && fracY <= y2) TR = TR + v;
                        if (fracX <= x2 && fracY >  y2) BL = BL + v;
                        if (fracX >  x2 && fracY >  y2) BR = BR + v;
                    } else {
                        const float wgt = fpx[(int64_t)k * s3.c];
                        if (fracX <= x2 && fracY <= y2) TL = fmaf(v, wgt, TL);
                        if (fracX >  x2 && fracY <= y2) TR = fmaf(v, wgt, TR);
                        if (fracX <= x2 && fracY >  y2) BL = fmaf(v, wgt, BL);
                        if (fracX >  x2 && fracY >  y2) BR = fmaf(v, wgt, BR);
                    }
                }
            }
        }
        dst[(int64_t)c * s1.c] = blend4(alpha, beta, TL, TR, BL, BR);
    }
}

// ------------------------------------------------------------------ backward, deformable variants

// filterinterpolation_cuda_kernel.cu:430-1215 (VARIANT 0), :1500-1935 (1), :2195-2567 (2).  Per valid
// pixel and tap: quadrant weight by integer index (0) or displaced position (1, 2); image gradient
// scattered to the clamped UNDISPLACED tap (the reference's own approximation) with atomics; filter
// and offset-field gradients belong to this pixel alone (plain read-modify-write, the same values as
// the reference's atomicAdd); flow gradient from the forward's quadrant sums.
// STAGED (round 3): the image-gradient addends of a 64x4 block are summed in a window of 64-bit LDS cells over the bounding
// box of the block's clamped taps, three channels at a time, and leave with one global atomic per non-zero cell (exact
// integer sums: the same bits as the per-tap scatter; see fi_backward_ori4_lds).  A block whose window does not fit, and
// every block of a call with non-finite inputs, raises its flag and returns; the !STAGED instance, launched afterwards with
// the flags, does those blocks only.
#define FD_CH 3
#define FD_CELLS 4608                               // 64-bit cells of LDS (36,864 bytes)
// FS = 4: the filter size at compile time -- offsets, weights and the filter / offset gradient sums of the pixel live in
// registers across the channel loop (summed in the same order from the cells' starting values: the same bits); FS = 0: any
// filter size, the cells are read and written in global memory per channel and tap.
template <int VARIANT, bool STAGED, int FS>
__global__ __launch_bounds__(VFI_TX * VFI_TY) void fi_backward_defor(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    const float* __restrict__ in4, const float* __restrict__ gout, unsigned long long* __restrict__ acc,
    const int* __restrict__ hdr, int* __restrict__ tileflag, float* g1, float* g2, float* g3, float* g4,
    int channel, int h, int w, int fs_arg, vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4) {
    const int fs = FS ? FS : fs_arg;
    constexpr int NA = FS ? FS * FS : 1;
    __shared__ unsigned long long cells[STAGED ? FD_CELLS : 1];
    __shared__ int box[4];
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    const int b = blockIdx.z;
    const int tile = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int tid = threadIdx.y * VFI_TX + threadIdx.x;
    if constexpr (!STAGED) {
        if (tileflag && !tileflag[tile]) return;
        if (x >= w || y >= h) return;
    }
    const bool inimg = x < w && y < h;
    const GradAccCtx gctx = gradacc_ctx(hdr);
    float fx = 0.0f, fy = 0.0f;
    if (inimg) {
        const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        fx = flow[0];
        fy = flow[s2.c];
    }
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    const bool valid = inimg && fi_valid(fx, fy, x2, y2, w, h);
    if constexpr (!STAGED) { if (!valid) return; }
    const int fs2 = fs * fs;
    const int ix = valid ? (int)x2 : 0, iy = valid ? (int)y2 : 0;
    const int L = ix + 1 - fs / 2, T = iy + 1 - fs / 2;
    int bx0 = 0, by0 = 0, bw = 0, n = 0;
    if constexpr (STAGED) {
        if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
        __syncthreads();
        const int wx0 = wave_min_i32(valid ? clampi(L, 0, w - 1) : INT_MAX), wy0 = wave_min_i32(valid ? clampi(T, 0, h - 1) : INT_MAX);
        const int wx1 = wave_max_i32(valid ? clampi(L + fs - 1, 0, w - 1) : INT_MIN), wy1 = wave_max_i32(valid ? clampi(T + fs - 1, 0, h - 1) : INT_MIN);
        if ((tid & 63) == 0 && wx0 != INT_MAX) {
            atomicMin(&box[0], wx0); atomicMin(&box[1], wy0);
            atomicMax(&box[2], wx1); atomicMax(&box[3], wy1);
        }
        __syncthreads();
        if (box[0] == INT_MAX) return;                      // (block-uniform: no pixel of the block has a gradient)
        bx0 = box[0]; by0 = box[1]; bw = box[2] - box[0] + 1;
        n = bw * (box[3] - box[1] + 1);
        if (gctx.nonfinite || n * min(FD_CH, channel) > FD_CELLS) {      // (block-uniform) left to the !STAGED instance
            if (tid == 0) tileflag[tile] = 1;
            return;
        }
    }
    const float alpha = x2 - (float)ix;
    const float beta = y2 - (float)iy;
    const float* img = in1 + (int64_t)b * s1.b;
    unsigned long long* gimg = acc + (int64_t)b * channel * h * w;       // dense [b][c][y][x] fixed-point sums
    const float* gpx = gout + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    // VARIANT 2: the third input IS the offset field and g3 its gradient; no filter
    const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    float* gfpx = g3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    const float* opx = (VARIANT == VFI_DEFOR_NOFILTER) ? fpx : in4 + (int64_t)b * s4.b + (int64_t)y * s4.h + x;
    float* gopx = (VARIANT == VFI_DEFOR_NOFILTER) ? gfpx : g4 + (int64_t)b * s4.b + (int64_t)y * s4.h + x;
    const int64_t ocs = (VARIANT == VFI_DEFOR_NOFILTER) ? s3.c : s4.c;
    const float kq[4] = { (1.0f - alpha) * (1.0f - beta), alpha * (1.0f - beta), (1.0f - alpha) * beta, alpha * beta };
    float offy[NA], offx[NA], wgts[NA], gfa[NA], goy[NA], gox[NA];
    if constexpr (FS != 0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            offy[k] = valid ? opx[(int64_t)k * ocs] : 0.0f;
            offx[k] = valid ? opx[(int64_t)(fs2 + k) * ocs] : 0.0f;
            goy[k] = valid ? gopx[(int64_t)k * ocs] : 0.0f;
            gox[k] = valid ? gopx[(int64_t)(fs2 + k) * ocs] : 0.0f;
            if constexpr (VARIANT != VFI_DEFOR_NOFILTER) {
                wgts[k] = valid ? fpx[(int64_t)k * s3.c] : 0.0f;
                gfa[k] = valid ? gfpx[(int64_t)k * s3.c] : 0.0f;
            }
        }
    }
    float gx = 0.0f, gy = 0.0f;
    const int chunk = STAGED ? FD_CH : channel;
    for (int c0 = 0; c0 < channel; c0 += chunk) {
    const int cn = min(chunk, channel - c0);
    if constexpr (STAGED) {
        for (int e = tid; e < n * cn; e += VFI_TX * VFI_TY) cells[e] = 0ull;
        __syncthreads();
    }
    for (int c = c0; c < c0 + cn && valid; ++c) {
        const float* p = img + (int64_t)c * s1.c;
        unsigned long long* gp = gimg + (int64_t)c * h * w;
        unsigned long long* win = cells + (STAGED ? (c - c0) * n : 0);
        float* gfp = g1 + (int64_t)b * s1.b + (int64_t)c * s1.c;    // (the fp32 scatter of a call with non-finite inputs)
        const float g = gpx[(int64_t)c * s1.c];
        const float qg[4] = { g * (1.0f - alpha) * (1.0f - beta), g * alpha * (1.0f - beta),
                              g * (1.0f - alpha) * beta,          g * alpha * beta };
        float q[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
        for (int dj = 0; dj < fs; ++dj) {
            const int j = T + dj, cj = clampi(j, 0, h - 1);
#pragma unroll
            for (int di = 0; di < fs; ++di) {
                const int i = L + di, ci = clampi(i, 0, w - 1);
                const int k = dj * fs + di;
                const int ka = FS ? k : 0;
                const float fracY = (float)cj + (FS ? offy[ka] : opx[(int64_t)k * ocs]);
                const float fracX = (float)ci + (FS ? offx[ka] : opx[(int64_t)(fs2 + k) * ocs]);
                int quad;
                if constexpr (VARIANT == VFI_DEFOR_OFFSET) quad = (j > iy ? 2 : 0) + (i > ix ? 1 : 0);
                else if (fracX <= x2 && fracY <= y2) quad = 0;
                else if (fracX > x2 && fracY <= y2) quad = 1;
                else if (fracX <= x2 && fracY > y2) quad = 2;
                else if (fracX > x2 && fracY > y2) quad = 3;
                else continue;                                      // NaN position: no quadrant
                const int Top = (int)fracY, Left = (int)fracX;
                const float phiY = fracY - (float)Top, phiX = fracX - (float)Left;
                const int64_t t = (int64_t)clampi(Top, 0, h - 1) * s1.h, bo = (int64_t)clampi(Top + 1, 0, h - 1) * s1.h;
                const int l = clampi(Left, 0, w - 1), r = clampi(Left + 1, 0, w - 1);
                const float vTL = p[t + l], vTR = p[t + r], vBL = p[bo + l], vBR = p[bo + r];
                float v = ((1.0f - phiX) * (1.0f - phiY)) * vTL;
                v = fmaf(phiX * (1.0f - phiY), vTR, v);
                v = fmaf((1.0f - phiX) * phiY, vBL, v);
                v = fmaf(phiY * phiX, vBR, v);
                float dY = (-(1.0f - phiX)) * vTL;
                dY = fmaf(1.0f - phiX, vBL, dY);
                dY = fmaf(-phiX, vTR, dY);
                dY = fmaf(phiX, vBR, dY);
                float dX = (-(1.0f - phiY)) * vTL;
                dX = fmaf(1.0f - phiY, vTR, dX);
                dX = fmaf(-phiY, vBL, dX);
                dX = fmaf(phiY, vBR, dX);
                const int64_t o = (int64_t)cj * w + ci;
                if constexpr (VARIANT == VFI_DEFOR_NOFILTER) {
                    if constexpr (STAGED) atomicAdd(&win[(cj - by0) * bw + ci - bx0], (unsigned long long)__float2ll_rn(qg[quad] * gctx.scale));
                    else gradacc_add(gp, gfp, o, (int64_t)cj * s1.h + ci, qg[quad], gctx);
                    q[quad] = q[quad] + v;
                    if constexpr (FS != 0) {
                        goy[ka] += g * kq[quad] * dY;
                        gox[ka] += g * kq[quad] * dX;
                    } else {
                        gopx[(int64_t)k * ocs] += g * kq[quad] * dY;
                        gopx[(int64_t)(fs2 + k) * ocs] += g * kq[quad] * dX;
                    }
                } else {
                    const float wgt = FS ? wgts[ka] : fpx[(int64_t)k * s3.c];
                    if constexpr (STAGED) atomicAdd(&win[(cj - by0) * bw + ci - bx0], (unsigned long long)__float2ll_rn(qg[quad] * wgt * gctx.scale));
                    else gradacc_add(gp, gfp, o, (int64_t)cj * s1.h + ci, qg[quad] * wgt, gctx);
                    q[quad] = fmaf(v, wgt, q[quad]);
                    if constexpr (FS != 0) {
                        gfa[ka] += qg[quad] * v;
                        goy[ka] += g * kq[quad] * dY * wgt;
                        gox[ka] += g * kq[quad] * dX * wgt;
                    } else {
                        gfpx[(int64_t)k * s3.c] += qg[quad] * v;
                        gopx[(int64_t)k * ocs] += g * kq[quad] * dY * wgt;
                        gopx[(int64_t)(fs2 + k) * ocs] += g * kq[quad] * dX * wgt;
                    }
                }
            }
        }
        {
            const float gamma = 1.0f - beta;
            float temp = gamma * (q[1] - q[0]);
            temp = fmaf(1.0f - gamma, q[3] - q[2], temp);
            gx = fmaf(g, temp, gx);
        }
        {
            const float gamma = 1.0f - alpha;
            float temp = gamma * (q[2] - q[0]);
            temp = fmaf(1.0f - gamma, q[3] - q[1], temp);
            gy = fmaf(g, temp, gy);
        }
    }
    if constexpr (STAGED) {
        __syncthreads();
        for (int e = tid; e < n * cn; e += VFI_TX * VFI_TY) {
            const unsigned long long v = cells[e];
            if (v != 0ull) {
                const int cc = e / n, r = e - cc * n;
                const int cy = r / bw, cx = r - cy * bw;
                atomicAdd(&gimg[(int64_t)(c0 + cc) * h * w + (int64_t)(by0 + cy) * w + bx0 + cx], v);
            }
        }
        __syncthreads();                                    // (the next pass zeroes the cells)
    }
    }
    if (valid) {
        float* gf = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        gf[0] = gx;
        gf[s2.c] = gy;
        if constexpr (FS != 0) {
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                gopx[(int64_t)k * ocs] = goy[k];
                gopx[(int64_t)(fs2 + k) * ocs] = gox[k];
                if constexpr (VARIANT != VFI_DEFOR_NOFILTER) gfpx[(int64_t)k * s3.c] = gfa[k];
            }
        }
    }
}

}  // namespace vfi

using namespace vfi;

// defined in filterinterp_lds.hip; returns VFI_OK / VFI_ERR_LAUNCH, or -1 when it declines the shape
extern "C" int vfi_filterinterp_forward_ori_lds_n(const float* input1, const float* input2, const float* input3,
                                                   float* output, int batch, int channel, int h, int w, int fs,
                                                   vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                                   vfi_stream_t stream);
extern "C" int vfi_filterinterp_forward_ori_lds(const float*, const float*, const float*, float*,
                                                 int, int, int, int, vfi_strides, vfi_strides, vfi_strides,
                                                 vfi_stream_t);

static int fi_filter_size(int filter_channels) { return (int)sqrtf((float)filter_channels); }

extern "C" int vfi_filterinterp_forward_ori_direct(const float* input1, const float* input2, const float* input3,
                                                    float* output, int batch, int channel, int h, int w,
                                                    int filter_channels, vfi_strides s1, vfi_strides s2,
                                                    vfi_strides s3, vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_channels <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !output) return VFI_ERR_SHAPE;
    const int fs = fi_filter_size(filter_channels);
    const dim3 grid = pixel_grid(w, h, batch), block(VFI_TX, VFI_TY, 1);
    hipStream_t st = (hipStream_t)stream;
    if (fs == 4)
        hipLaunchKernelGGL(fi_forward_ori_direct<true>, grid, block, 0, st, input1, input2, input3, output,
                           channel, h, w, fs, s1, s2, s3);
    else
        hipLaunchKernelGGL(fi_forward_ori_direct<false>, grid, block, 0, st, input1, input2, input3, output,
                           channel, h, w, fs, s1, s2, s3);
    return launch_status();
}

extern "C" int vfi_filterinterp_forward_ori(const float* input1, const float* input2, const float* input3,
                                             float* output, int batch, int channel, int h, int w,
                                             int filter_channels, vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                             vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_channels <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !output) return VFI_ERR_SHAPE;
    const int fs = fi_filter_size(filter_channels);
    if (fs == 4) {
        const int r = vfi_filterinterp_forward_ori_lds(input1, input2, input3, output, batch, channel, h, w,
                                                       s1, s2, s3, stream);
        if (r != -1) return r;
    } else if (fs * fs == filter_channels) {
        const int r = vfi_filterinterp_forward_ori_lds_n(input1, input2, input3, output, batch, channel, h, w, fs,
                                                         s1, s2, s3, stream);
        if (r != -1) return r;
    }
    return vfi_filterinterp_forward_ori_direct(input1, input2, input3, output, batch, channel, h, w,
                                               filter_channels, s1, s2, s3, stream);
}

extern "C" int vfi_filterinterp_backward_ori(const float* input1, const float* input2, const float* input3,
                                              const float* gradoutput, float* gradinput1, float* gradinput2,
                                              float* gradinput3, int batch, int channel, int h, int w,
                                              int filter_channels, vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                              vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_channels <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !gradoutput || !gradinput1 || !gradinput2 || !gradinput3)
        return VFI_ERR_SHAPE;
    const int fs = fi_filter_size(filter_channels);
    unsigned long long* acc;
    int* hdr;
    int* flags = nullptr;                                   // one word per FB_TW x FB_TH (64 x 8) tile: "the staged kernel left it alone"
    const int ntiles = ((w + FB_TW - 1) / FB_TW) * ((h + FB_TH - 1) / FB_TH) * batch;
    int err = gradacc_begin((hipStream_t)stream, gradoutput, batch, channel, h, w, s1, input3, filter_channels, s3, &acc, &hdr,
                            fs == 4 ? ntiles : 0, &flags);
    if (err != VFI_OK) return err;
    static_assert(VFI_TX == FB_TW && FB_TH % VFI_TY == 0, "fi_backward_ori's blocks nest in the staged kernel's tiles");
    if (fs == 4)
        hipLaunchKernelGGL(fi_backward_ori4_lds, dim3((w + FB_TW - 1) / FB_TW, (h + FB_TH - 1) / FB_TH, batch), dim3(FB_THREADS), 0,
                           (hipStream_t)stream, input1, input2, input3, gradoutput, acc, hdr, flags, gradinput2, gradinput3,
                           channel, h, w, s1, s2, s3);
    hipLaunchKernelGGL(fi_backward_ori, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input1, input2, input3, gradoutput, acc, hdr, gradinput1, gradinput2, gradinput3,
                       channel, h, w, fs, s1, s2, s3, fs == 4 ? flags : nullptr);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    return gradacc_finish((hipStream_t)stream, acc, hdr, gradinput1, batch, channel, h, w, s1);
}

extern "C" int vfi_filterinterp_forward_defor_lds(int variant, const float* input1, const float* input2,
                                                   const float* input3, const float* input4, float* output,
                                                   int batch, int channel, int h, int w, int filter_size,
                                                   vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4,
                                                   vfi_stream_t stream);

static int defor_forward(bool allow_staged, int variant, const float* input1, const float* input2,
                         const float* input3, const float* input4, float* output,
                         int batch, int channel, int h, int w, int filter_size,
                         vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4,
                         vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_size <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !output) return VFI_ERR_SHAPE;
    if (variant != VFI_DEFOR_NOFILTER && !input4) return VFI_ERR_SHAPE;
    const dim3 grid = pixel_grid(w, h, batch), block(VFI_TX, VFI_TY, 1);
    hipStream_t st = (hipStream_t)stream;
    if ((filter_size == 4 || filter_size == 6) && allow_staged && variant >= 0 && variant <= 2) {
        // LDS-staged kernel (filterinterp_defor_lds.hip); -1 = not applicable
        const int err = vfi_filterinterp_forward_defor_lds(variant, input1, input2, input3, input4, output, batch,
                                                           channel, h, w, filter_size, s1, s2, s3, s4, stream);
        if (err >= 0) return err;
    }
    switch (variant) {
    case VFI_DEFOR_OFFSET:
        // the reference kernel has a body for fs 4 and 6 only; other sizes leave
        // the caller's (zero-filled) output untouched (:68)
        if (!(filter_size == 4 || filter_size == 6)) return VFI_OK;
        hipLaunchKernelGGL(fi_forward_defor<VFI_DEFOR_OFFSET>, grid, block, 0, st, input1, input2, input3, input4,
                           output, channel, h, w, filter_size, s1, s2, s3, s4);
        break;
    case VFI_DEFOR_REGION:
        hipLaunchKernelGGL(fi_forward_defor<VFI_DEFOR_REGION>, grid, block, 0, st, input1, input2, input3, input4,
                           output, channel, h, w, filter_size, s1, s2, s3, s4);
        break;
    case VFI_DEFOR_NOFILTER:
        hipLaunchKernelGGL(fi_forward_defor<VFI_DEFOR_NOFILTER>, grid, block, 0, st, input1, input2, input3,
                           input3, output, channel, h, w, filter_size, s1, s2, s3, s3);
        break;
    default:
        return VFI_ERR_SHAPE;
    }
    return launch_status();
}

extern "C" int vfi_filterinterp_forward_defor(int variant, const float* input1, const float* input2,
                                               const float* input3, const float* input4, float* output,
                                               int batch, int channel, int h, int w, int filter_size,
                                               vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4,
                                               vfi_stream_t stream) {
    return defor_forward(true, variant, input1, input2, input3, input4, output, batch, channel, h, w, filter_size, s1, s2, s3,
                         s4, stream);
}

// internal (not in vfi_hip.h): always the general one-thread-per-pixel kernel -- the tests compare the staged kernel with it
extern "C" int vfi_filterinterp_forward_defor_general(int variant, const float* input1, const float* input2,
                                                       const float* input3, const float* input4, float* output,
                                                       int batch, int channel, int h, int w, int filter_size,
                                                       vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4,
                                                       vfi_stream_t stream) {
    return defor_forward(false, variant, input1, input2, input3, input4, output, batch, channel, h, w, filter_size, s1, s2, s3,
                         s4, stream);
}

// defined in filterinterp_defor_bwd_lds.hip
extern "C" int vfi_filterinterp_backward_defor_lds(int variant, const float* input1, const float* input2, const float* input3,
                                                    const float* input4, const float* gradoutput, unsigned long long* acc,
                                                    const int* hdr, int* flags, float* gradinput2, float* gradinput3,
                                                    float* gradinput4, int batch, int channel, int h, int w, vfi_strides s1,
                                                    vfi_strides s2, vfi_strides s3, vfi_strides s4, vfi_stream_t stream);

extern "C" int vfi_filterinterp_backward_defor(int variant, const float* input1, const float* input2,
                                                const float* input3, const float* input4, const float* gradoutput,
                                                float* gradinput1, float* gradinput2, float* gradinput3,
                                                float* gradinput4, int batch, int channel, int h, int w,
                                                int filter_size, vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                                vfi_strides s4, vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_size <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !gradoutput || !gradinput1 || !gradinput2 || !gradinput3) return VFI_ERR_SHAPE;
    if (variant != VFI_DEFOR_NOFILTER && (!input4 || !gradinput4)) return VFI_ERR_SHAPE;
    if (variant < 0 || variant > 2) return VFI_ERR_SHAPE;
    const dim3 grid = pixel_grid(w, h, batch), block(VFI_TX, VFI_TY, 1);
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* acc;
    int* hdr;
    // (the tap weights of the variant without a filter are 1)
    int* flags = nullptr;                                   // one word per block: "the staged instance left it alone"
    const int err = gradacc_begin(st, gradoutput, batch, channel, h, w, s1, variant == VFI_DEFOR_NOFILTER ? nullptr : input3,
                                  filter_size * filter_size, s3, &acc, &hdr, (int)(grid.x * grid.y * grid.z), &flags);
    if (err != VFI_OK) return err;
#define FD_LAUNCH2(V, F, I3, I4, G3, G4, S4) \
        hipLaunchKernelGGL((fi_backward_defor<V, true, F>), grid, block, 0, st, input1, input2, I3, I4, gradoutput, acc, hdr, flags, \
                           gradinput1, gradinput2, G3, G4, channel, h, w, filter_size, s1, s2, s3, S4); \
        hipLaunchKernelGGL((fi_backward_defor<V, false, F>), grid, block, 0, st, input1, input2, I3, I4, gradoutput, acc, hdr, flags, \
                           gradinput1, gradinput2, G3, G4, channel, h, w, filter_size, s1, s2, s3, S4)
    // fs == 4: the LDS-staged kernel (filterinterp_defor_bwd_lds.hip), then the per-tap instance for the blocks it flagged
#define FD_LAUNCH(V, I3, I4, G3, G4, S4) \
        if (staged4 == 0) hipLaunchKernelGGL((fi_backward_defor<V, false, 4>), grid, block, 0, st, input1, input2, I3, I4, gradoutput, acc, hdr, flags, \
                                             gradinput1, gradinput2, G3, G4, channel, h, w, filter_size, s1, s2, s3, S4); \
        else { FD_LAUNCH2(V, 0, I3, I4, G3, G4, S4); }
    int staged4 = -1;
    if (filter_size == 4) {
        staged4 = vfi_filterinterp_backward_defor_lds(variant, input1, input2, input3, input4, gradoutput, acc, hdr, flags, gradinput2,
                                                      gradinput3, gradinput4, batch, channel, h, w, s1, s2, s3, s4, stream);
        if (staged4 != 0 && staged4 != -1) return VFI_ERR_LAUNCH;
    }
    switch (variant) {
    case VFI_DEFOR_OFFSET: FD_LAUNCH(VFI_DEFOR_OFFSET, input3, input4, gradinput3, gradinput4, s4); break;
    case VFI_DEFOR_REGION: FD_LAUNCH(VFI_DEFOR_REGION, input3, input4, gradinput3, gradinput4, s4); break;
    default:               FD_LAUNCH(VFI_DEFOR_NOFILTER, input3, input3, gradinput3, gradinput3, s3); break;
    }
#undef FD_LAUNCH
#undef FD_LAUNCH2
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    return gradacc_finish(st, acc, hdr, gradinput1, batch, channel, h, w, s1);
}
