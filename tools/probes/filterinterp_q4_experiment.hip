// filterinterp_q4_experiment.hip -- NOT BUILT.  Round-3 experiment, measured and rejected: FilterInterpolation (_ori, fs == 4)
// forward for frames (C <= 4) with a thread owning four consecutive pixels (16-byte flow / filter loads, 16-byte LDS-DMA,
// 16-byte stores).  Bit-exact with the staged kernel (tools/fi_soak.py, 400 + 300 + 200 cases), but at 1152x1984, C=3 it took
// 81.6 us with 4-byte tap reads (the quads of a wave sit 16 bytes apart: 4-way LDS bank conflicts) and 88.5 us with two
// aligned 16-byte tap reads + selects per tap row, against 41 us for filterinterp_lds.hip's layout (a wave = 64 consecutive
// pixels of one row).  PMC (tools/pmc_fi3.sh): 1167 VALU + 449 SALU instructions per wave, 52 % of the wave cycles in
// SQ_WAIT_INST_ANY, 31 % in SQ_WAIT_ANY -- the four-pixel thread serialises what the two-pixel layout spreads over twice
// the waves.  Kept as the record of the experiment; to try it again, add it to csrc/Makefile and call
// vfi_filterinterp_forward_ori_q4 at the top of forward_ori_lds (filterinterp_lds.hip).
//
// Semantics: filterinterpolation_cuda_kernel.cu:2692-2823, bit for bit the staged kernel's of filterinterp_lds.hip
// (same taps, same order: fi4_pixel).
//
// Why a kernel of its own: with three channels the op IS its prologue -- per pixel 2 flow + 16 filter floats in, 3 image
// floats in and 3 out: 219 MB at 1080p, 146 MB of them the filter planes -- and the 196-channel kernel reads them with
// 4-byte lanes, 36 loads per thread (41 us hot / 51 us with cold caches: 4.3 TB/s).  A walk of the planes with 16-byte
// lanes reaches 5.3 TB/s on this chip against ~4 with 4-byte lanes (tools/probes/tile_walk_probe*.hip).  Here a thread
// owns FOUR consecutive pixels of a row: the flow and the 16 filter planes arrive as 18 loads of 16 bytes, the results
// leave as one 16-byte store per channel; a workgroup of 256 threads owns a 64x16 tile, four workgroups per CU.
// All channels' windows are staged at once (no ring: there is nothing to pipeline over three channels) -- by 16-byte
// LDS-DMA when the window lies inside the image (its left edge moved to a multiple of four pixels), by the 4-byte DMA
// with replicated borders otherwise -- then every tap is an LDS read at an immediate offset, as in the big kernel.
// A tile whose windows do not fit (wildly divergent flow) gathers from global memory, decided per workgroup.
#include "filterinterp_dev.h"

#include <limits.h>

namespace vfi {

#define Q4_TW 64
#define Q4_TH 16
#define Q4_THREADS 256
#define Q4_HDR 16                                   // floats at the head of the LDS array (bounding box)
#define Q4_WIN 9968                                 // window floats, all channels: with the header 39,936 B = 32 LDS granules, 4 workgroups per CU
#define Q4_MAXC 4
#define Q4_XCDS 8

typedef __attribute__((address_space(3))) void* q4_lptr_t;
typedef float q4_v4f __attribute__((ext_vector_type(4)));
typedef int q4_v4i __attribute__((ext_vector_type(4)));

struct Q4Blend { const float* other; float* out; float w0, w2; };

__device__ __forceinline__ q4_v4f q4_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(q4_v4f, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <bool BLEND>
__global__ __launch_bounds__(Q4_THREADS, 4) void fi_forward_ori_q4(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3, float* __restrict__ out,
    int channel, int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides s3, int tiles_x, int tiles_y, int ntiles, Q4Blend blend) {
    // ONE LDS array (a second __shared__ object beside an LDS-DMA target makes hipcc drain vmcnt before LDS reads)
    __shared__ __attribute__((aligned(16))) float lds[Q4_HDR + Q4_WIN];
    int* box = reinterpret_cast<int*>(lds);
    float* win = lds + Q4_HDR;

    // four horizontally consecutive tiles on one XCD (their windows share 128-byte lines): see filterinterp_lds.hip
    const int bid = blockIdx.x;
    const int xs = bid % Q4_XCDS, kk = bid / Q4_XCDS;
    const int tile = ((kk >> 2) * Q4_XCDS + xs) * 4 + (kk & 3);
    if (tile >= ntiles) return;
    const int b = tile / (tiles_x * tiles_y);
    const int trem = tile - b * (tiles_x * tiles_y);
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;

    const int tid = threadIdx.x;
    const int q = tid & 15, r = tid >> 4;
    const int x0 = txi * Q4_TW + 4 * q, y = tyi * Q4_TH + r;
    const bool inimg = y < h && x0 < w;                     // (w is a multiple of four: a quad is inside or outside)

    // ---- flow and the 16 filter planes of the thread's four pixels: 18 loads of 16 bytes, all in flight at once
    // (a thread outside the image gets an out-of-range offset: zeros, no traffic)
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(in2 + (int64_t)b * s2.b), 0,
                                                                         (int)(4 * (s2.c + (int64_t)(h - 1) * s2.h + w)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc((void*)(in3 + (int64_t)b * s3.b), 0,
                                                                         (int)(4 * (15 * s3.c + (int64_t)(h - 1) * s3.h + w)), 0x00020000);
    const unsigned vo2 = inimg ? 4u * (unsigned)(y * (int)s2.h + x0) : 0x80000000u;
    const unsigned vo3 = inimg ? 4u * (unsigned)(y * (int)s3.h + x0) : 0x80000000u;
    const q4_v4f fx4 = q4_load(rs2, vo2, 0u), fy4 = q4_load(rs2, vo2, 4u * (unsigned)s2.c);
    q4_v4f F[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) F[k] = q4_load(rs3, vo3, 4u * (unsigned)(k * s3.c));

    // ---- the four pixels: validity, window origin, blend weights (:2735-2748)
    bool valid[4];
    int L[4], T[4];
    float alpha[4], beta[4];
    int bx_lo = INT_MAX, by_lo = INT_MAX, bx_hi = INT_MIN, by_hi = INT_MIN;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = x0 + j;
        const float fx = fx4[j], fy = fy4[j];
        const float x2 = (float)x + fx, y2 = (float)y + fy;
        valid[j] = inimg && fi_valid(fx, fy, x2, y2, w, h);
        const int ix = valid[j] ? (int)x2 : 0, iy = valid[j] ? (int)y2 : 0;
        L[j] = ix - 1;                                      // ix + 1 - fs/2, fs == 4
        T[j] = iy - 1;
        alpha[j] = x2 - (float)ix;
        beta[j] = y2 - (float)iy;
        if (valid[j]) {
            bx_lo = min(bx_lo, L[j]); by_lo = min(by_lo, T[j]);
            bx_hi = max(bx_hi, L[j] + 3); by_hi = max(by_hi, T[j] + 3);
        }
    }

    // ---- bounding box of every tap of the tile (unclamped window coordinates)
    if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
    __syncthreads();
    {
        const int wx0 = wave_min_i32(bx_lo), wy0 = wave_min_i32(by_lo);
        const int wx1 = wave_max_i32(bx_hi), wy1 = wave_max_i32(by_hi);
        if ((tid & 63) == 0 && wx0 != INT_MAX) {
            atomicMin(&box[0], wx0); atomicMin(&box[1], wy0);
            atomicMax(&box[2], wx1); atomicMax(&box[3], wy1);
        }
    }
    __syncthreads();
    const bool any_valid = box[0] != INT_MAX;
    const int by0 = box[1];
    const int bh = any_valid ? box[3] - box[1] + 1 : 0;
    // The window's left edge sits at a multiple of four pixels (also when it is left of the image: & ~3 floors): a tap row is
    // then read as TWO ALIGNED 16-byte LDS reads and the four taps are picked by the column's low bits (below) -- the quads
    // of a wave's lanes are 16 bytes apart, so aligned 16-byte reads are conflict-free where 4-byte reads at a stride of
    // four floats are 4-way conflicts (81 us instead of 41 for the whole launch).  A window inside the image needs no
    // border replication and is staged by 16-byte DMA.
    const bool clean = any_valid && box[0] >= 0 && box[1] >= 0 && box[2] < w && box[3] < h;
    const int bx0 = box[0] & ~3;
    const int bw = any_valid ? box[2] - bx0 + 1 : 0;
    const int pitch = (bw + 4 + 31) & ~31;                  // the second read of a row may reach four floats past the last tap
    const int n = pitch * bh;                               // floats per channel

    const float* img = in1 + (int64_t)b * s1.b;
    float* dst = out + (int64_t)b * s1.b;
    const int hs = (int)s1.h;
    const int64_t cs = s1.c;
    const unsigned pix = (unsigned)(y * hs + x0);
    const int plane_bytes = 4 * ((h - 1) * hs + w);

    if (n * channel > Q4_WIN) {
        // windows too large for the LDS: gather from global memory (workgroup-uniform branch), one pixel at a time; the
        // filter taps are fetched again per pixel (keeping the 64 registers alive through this branch spills the main path)
        if (!inimg) return;
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            const float fx = in2[(int64_t)b * s2.b + (int64_t)y * s2.h + x], fy = in2[(int64_t)b * s2.b + s2.c + (int64_t)y * s2.h + x];
            const float x2 = (float)x + fx, y2 = (float)y + fy;
            float* o = dst + pix + j;
            if (fi_valid(fx, fy, x2, y2, w, h)) {
                const int ix = (int)x2, iy = (int)y2;
                float f[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) f[k] = in3[(int64_t)b * s3.b + (int64_t)k * s3.c + (int64_t)y * s3.h + x];
                fi4_channels_direct(img, o, 0, channel, cs, hs, h, w, ix - 1, iy - 1, f, x2 - (float)ix, y2 - (float)iy);
            } else {
                for (int c = 0; c < channel; ++c) o[(int64_t)c * cs] = img[(int64_t)c * cs + pix + j];
            }
            if constexpr (BLEND)                                // (this thread wrote dst[...] itself: it reads its own stores)
                for (int c = 0; c < channel; ++c) {
                    const float q0 = blend.other[(int64_t)b * s1.b + (int64_t)c * cs + pix + j] * blend.w0, q2 = o[(int64_t)c * cs] * blend.w2;
                    blend.out[(int64_t)b * s1.b + (int64_t)c * cs + pix + j] = q0 + q2;
                }
        }
        return;
    }

    // ---- stage every channel's window
    if (clean) {
        const int upr = pitch >> 2;                         // 16-byte units per window row
        const int uused = (bw + 3) >> 2;                    // ... that hold pixels (the window ends inside the image, w % 4 == 0)
        const int nunits = upr * bh;
        const float inv_upr8 = 1.0f / (float)(upr >> 3);    // (upr is a multiple of 8)
        for (int e = tid; e < nunits; e += Q4_THREADS) {
            const int row = (int)(((float)(e >> 3) + 0.5f) * inv_upr8);     // e / upr (see fi_row_of)
            const int cu = e - row * upr;
            const unsigned off = cu < uused ? 4u * (unsigned)((by0 + row) * hs + bx0 + 4 * cu) : 0x80000000u;
            for (int c = 0; c < channel; ++c) {
                const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)c * cs), 0, plane_bytes, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (q4_lptr_t)(win + c * n + 4 * e), 16, off, 0, 0, 0);
            }
        }
    } else {
        const float inv_pitch32 = 1.0f / (float)(pitch >> 5);
        for (int e = tid; e < n; e += Q4_THREADS) {
            const int row = fi_row_of(e, inv_pitch32);
            const int col = e - row * pitch;
            const unsigned off = col < bw ? 4u * (unsigned)(clampi(by0 + row, 0, h - 1) * hs + clampi(bx0 + col, 0, w - 1)) : 0x80000000u;
            for (int c = 0; c < channel; ++c) {
                const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)c * cs), 0, plane_bytes, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (q4_lptr_t)(win + c * n + e), 4, off, 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the windows (and the filter planes) have landed ...
    __syncthreads();                                        // ... in every wave

    // ---- the taps: two aligned 16-byte LDS reads per row, the four taps selected by the low bits of the window column
    int lb[4];
    bool sh1[4], sh2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cl = valid[j] ? L[j] - bx0 : 0;
        lb[j] = valid[j] ? (T[j] - by0) * pitch + (cl & ~3) : 0;
        sh1[j] = (cl & 1) != 0;
        sh2[j] = (cl & 2) != 0;
    }
    if (!inimg) return;
    for (int c = 0; c < channel; ++c) {
        const float* wc = win + c * n;
        q4_v4f res;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float val;
            if (valid[j]) {
                // row by row, in fi4_pixel's order (rows outer, columns inner per quadrant; `acc += a*b` fused)
                const float* t = wc + lb[j];
                float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const q4_v4f a = *reinterpret_cast<const q4_v4f*>(t + rr * pitch), bq = *reinterpret_cast<const q4_v4f*>(t + rr * pitch + 4);
                    const bool s2 = sh2[j], s1b = sh1[j];
                    const float u0 = s2 ? a.z : a.x, u1 = s2 ? a.w : a.y, u2 = s2 ? bq.x : a.z, u3 = s2 ? bq.y : a.w, u4 = s2 ? bq.z : bq.x;
                    const float v[4] = {s1b ? u1 : u0, s1b ? u2 : u1, s1b ? u3 : u2, s1b ? u4 : u3};
                    const float f0 = F[rr * 4 + 0][j], f1 = F[rr * 4 + 1][j], f2 = F[rr * 4 + 2][j], f3 = F[rr * 4 + 3][j];
                    if (rr == 0) { TL = v[0] * f0; TL = fmaf(v[1], f1, TL); TR = v[2] * f2; TR = fmaf(v[3], f3, TR); }
                    if (rr == 1) { TL = fmaf(v[0], f0, TL); TL = fmaf(v[1], f1, TL); TR = fmaf(v[2], f2, TR); TR = fmaf(v[3], f3, TR); }
                    if (rr == 2) { BL = v[0] * f0; BL = fmaf(v[1], f1, BL); BR = v[2] * f2; BR = fmaf(v[3], f3, BR); }
                    if (rr == 3) { BL = fmaf(v[0], f0, BL); BL = fmaf(v[1], f1, BL); BR = fmaf(v[2], f2, BR); BR = fmaf(v[3], f3, BR); }
                }
                val = blend4(alpha[j], beta[j], TL, TR, BL, BR);
            } else {
                val = img[(int64_t)c * cs + pix + j];       // copy-through of the (rare) invalid pixels (:2814-2818)
            }
            res[j] = val;
        }
        *reinterpret_cast<q4_v4f*>(dst + (int64_t)c * cs + pix) = res;
        if constexpr (BLEND) {
            const q4_v4f o = *reinterpret_cast<const q4_v4f*>(blend.other + (int64_t)b * s1.b + (int64_t)c * cs + pix);
            q4_v4f bo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float q0 = o[j] * blend.w0, q2 = res[j] * blend.w2; bo[j] = q0 + q2; }
            *reinterpret_cast<q4_v4f*>(blend.out + (int64_t)b * s1.b + (int64_t)c * cs + pix) = bo;
        }
    }
}

}  // namespace vfi

using namespace vfi;

static bool q4_aligned(const void* p, const vfi_strides& s) {
    return (uintptr_t)p % 16 == 0 && s.b % 4 == 0 && s.c % 4 == 0 && s.h % 4 == 0;
}

// internal: returns -1 when this path does not apply (the caller takes the general staged kernel).  other / blend:
// the fused blend of DAIN.FilterInterpolate's second launch (nullptr: none); both have input1's strides.
extern "C" int vfi_filterinterp_forward_ori_q4(const float* input1, const float* input2, const float* input3, float* output,
                                                const float* other, float* blend, float w0, float w2, int batch, int channel,
                                                int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_stream_t stream) {
    if (channel > Q4_MAXC || (w & 3) != 0) return -1;
    if (!q4_aligned(input1, s1) || !q4_aligned(output, s1) || !q4_aligned(input2, s2) || !q4_aligned(input3, s3)) return -1;
    if ((other || blend) && (!other || !blend || (uintptr_t)other % 16 != 0 || (uintptr_t)blend % 16 != 0)) return -1;
    // 32-bit byte offsets inside a plane / inside the flow and filter tensors of one batch item
    if ((int64_t)h * s1.h * 4 > INT_MAX || 4 * (s2.c + (int64_t)(h - 1) * s2.h + w) > INT_MAX ||
        4 * (15 * s3.c + (int64_t)(h - 1) * s3.h + w) > INT_MAX)
        return -1;
    const int tiles_x = (w + Q4_TW - 1) / Q4_TW, tiles_y = (h + Q4_TH - 1) / Q4_TH;
    const int64_t nt = (int64_t)tiles_x * tiles_y * batch;
    if (nt > (1 << 28)) return -1;
    const int ntiles = (int)nt;
    const int grid = ((ntiles + 4 * Q4_XCDS - 1) / (4 * Q4_XCDS)) * (4 * Q4_XCDS);      // whole groups of 4 tiles x 8 XCDs
    const Q4Blend bl{other, blend, w0, w2};
    if (blend)
        hipLaunchKernelGGL((fi_forward_ori_q4<true>), dim3(grid), dim3(Q4_THREADS), 0, (hipStream_t)stream, input1, input2, input3,
                           output, channel, h, w, s1, s2, s3, tiles_x, tiles_y, ntiles, bl);
    else
        hipLaunchKernelGGL((fi_forward_ori_q4<false>), dim3(grid), dim3(Q4_THREADS), 0, (hipStream_t)stream, input1, input2, input3,
                           output, channel, h, w, s1, s2, s3, tiles_x, tiles_y, ntiles, bl);
    return launch_status();
}
