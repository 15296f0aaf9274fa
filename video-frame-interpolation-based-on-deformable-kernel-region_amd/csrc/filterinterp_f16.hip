// filterinterp_f16.hip -- FilterInterpolation (_ori forward) with fp16 STORAGE of the image and the
// output and fp32 arithmetic (BASELINE.json configs[2]: "fp16 storage / fp32 accum").
//
// Semantics: filterinterpolation_cuda_kernel.cu:2692-2823 applied to the fp16 values widened to
// fp32; flow and filter stay fp32; the fp32 result is rounded to fp16 (round to nearest even).
//
//   fi_forward_ori_direct_f16   one thread per pixel, any filter size, the reference's operation
//                               order: bit-exact with the fp32 oracle run on the widened inputs and
//                               rounded once.  Fallback of the staged kernel and its yardstick.
//   fi_forward_ori_lds_f16      fs == 4, the 196-channel case: the tiling, LDS-DMA ring and channel
//                               loop of fi_forward_ori_lds (filterinterp_lds.hip) with half the
//                               bytes everywhere -- a staged window element is one dword = two
//                               halves, a tap row is three dword LDS reads (instead of four) shifted
//                               into place by v_alignbit with a per-lane amount, and the products
//                               are v_fma_mix_f32 straight from the packed halves.
// Two things differ from the fp32 staged kernel, both inside the stated fp16 tolerance (2e-3 for
// images in [0, 1], tests/test_gpu_parity.py):
//   * the pixel is one 16-term dot product with weights filter x bilinear weight folded once per
//     tile (same products as the reference, one accumulation chain instead of four + blend);
//   * image columns are never replicated in LDS (a dword holds two pixels, so per-pixel clamping
//     of a dword load is impossible): a pixel whose taps leave the image left or right gets its
//     window moved inside and the weights of the clamped taps added onto the column they clamp to.
#define FI_PITCH_SKEW 16                            // (dwords; see fi_pitch_for)
#include "filterinterp_dev.h"

#include <hip/hip_fp16.h>
#include <limits.h>

namespace vfi {

#define F16_TW 64
#define F16_TH 16
#define F16_PX 2
#define F16_THREADS (F16_TW * F16_TH / F16_PX)
#define F16_PASS_ROWS (F16_TH / F16_PX)
#define F16_HDR 16
#define F16_RING_DWORDS 15984
#define F16_RMAX 5
#define F16_KTOP 8

typedef __attribute__((address_space(3))) void* f16_lptr_t;

// fp32 -> fp16, round to nearest even, of a value that has first been rounded to fp32.  The empty asm
// keeps hipcc from folding the conversion into the last FMA (v_fma_mixlo_f16 rounds the exact FMA
// result to fp16 once; "the fp32 op, then .half()" rounds twice, and ties can fall the other way).
__device__ __forceinline__ __half f16_store_value(float v) {
    asm volatile("" : "+v"(v));
    return __float2half_rn(v);
}

// ------------------------------------------------------------------ direct kernel (reference order)

__global__ __launch_bounds__(VFI_TX * VFI_TY) void fi_forward_ori_direct_f16(
    const __half* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    __half* __restrict__ out, int channel, int h, int w, int fs,
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
    const __half* img = in1 + (int64_t)b * s1.b;
    __half* dst = out + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    if (!fi_valid(fx, fy, x2, y2, w, h)) {
        const __half* src = img + (int64_t)y * s1.h + x;
        for (int c = 0; c < channel; ++c) dst[(int64_t)c * s1.c] = src[(int64_t)c * s1.c];
        return;
    }
    const int ix = (int)x2, iy = (int)y2;
    const int L = ix + 1 - fs / 2, T = iy + 1 - fs / 2;
    const int R = L + fs, Bm = T + fs;
    const float alpha = x2 - (float)ix;
    const float beta = y2 - (float)iy;
    const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    for (int c = 0; c < channel; ++c) {
        const __half* plane = img + (int64_t)c * s1.c;
        float q[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        // quadrant by quadrant, rows outer / columns inner (:2749-2787)
        for (int quad = 0; quad < 4; ++quad) {
            const int j0 = (quad & 2) ? iy + 1 : T, j1 = (quad & 2) ? Bm : iy + 1;
            const int i0 = (quad & 1) ? ix + 1 : L, i1 = (quad & 1) ? R : ix + 1;
            float acc = 0.0f;
            for (int j = j0; j < j1; ++j) {
                const __half* row = plane + (int64_t)clampi(j, 0, h - 1) * s1.h;
                for (int i = i0; i < i1; ++i)
                    acc = fmaf(__half2float(row[clampi(i, 0, w - 1)]), fpx[(int64_t)((j - T) * fs + (i - L)) * s3.c], acc);
            }
            q[quad] = acc;
        }
        dst[(int64_t)c * s1.c] = f16_store_value(blend4(alpha, beta, q[0], q[1], q[2], q[3]));
    }
}

// ------------------------------------------------------------------ LDS-staged kernel, fs == 4

struct F16Window { int bx0, by0, bw, bh, pitch, h, w, hs; };   // bx0 even; bw, pitch in dwords
struct F16Pixel {
    bool valid, inimg;
    int lbase;              // half index of the window origin inside a staged window (row pitch 2 * pitch)
    unsigned pix;
    float g[16];            // folded tap weights on the (possibly moved) 4x4 window
};

template <int K>
__device__ __forceinline__ void f16_wait_windows(int younger_groups) {
    switch (younger_groups) {
    case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
    case 2:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * K) : "memory"); break;
    }
}

// the two halves of a packed dword, widened (v_fma_mix_f32 reads them in place)
__device__ __forceinline__ float f16_lo(unsigned d) { return __half2float(__ushort_as_half((unsigned short)(d & 0xffffu))); }
__device__ __forceinline__ float f16_hi(unsigned d) { return __half2float(__ushort_as_half((unsigned short)(d >> 16))); }

// Channel loop of one workgroup, written like fi_run_channels_lean (filterinterp_lds.hip: what bounds this loop is the
// instruction count): ring geometry a compile-time function of K and one constant s_waitcnt in the steady state, running
// plane descriptors, M0 formed on the scalar unit, tap reads as asm with two lgkmcnt waits per channel and the second
// pixel's first rows in flight under the first pixel's arithmetic, range-checked buffer stores.
template <int K>
__device__ __forceinline__ void f16_run_channels(const __half* __restrict__ img, __half* __restrict__ out, int64_t cs,
                                                 int c_begin, int c_end, int tid, const F16Window& win,
                                                 const F16Pixel (&px)[F16_PX], unsigned* __restrict__ ring) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    static_assert(F16_PX == 2, "two pixels per thread");
    constexpr int NP = K * F16_THREADS;                     // dwords per ring slot
    constexpr int R = (F16_RING_DWORDS / NP) < F16_RMAX ? (F16_RING_DWORDS / NP) : F16_RMAX;
    constexpr int D = R - 1;
    static_assert(D >= 1 && (D - 1) * K <= 63, "ring geometry");
    if (c_begin >= c_end) return;
    // dword e = tid + k*F16_THREADS of the staged window, row-major, `pitch` dwords per row; rows
    // clamped to the image, columns never out of it; pad dwords get an out-of-range offset (the
    // load returns 0 without touching memory)
    const float inv_pitch32 = 1.0f / (float)win.pitch;
    unsigned goff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + k * F16_THREADS;
        const int r = fi_row_of(e, inv_pitch32);
        const int col = e - r * win.pitch;
        const unsigned off = 2u * (unsigned)(clampi(win.by0 + r, 0, win.h - 1) * win.hs + win.bx0 + 2 * col);
        goff[k] = (col < win.bw && r < win.bh) ? off : 0x80000000u;
    }
    const int plane_bytes = (2 * ((win.h - 1) * win.hs + win.w) + 3) & ~3;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
    const unsigned ring_lds = (unsigned)(uintptr_t)(f16_lptr_t)ring;
    const unsigned pitch4 = 4u * (unsigned)win.pitch;
    unsigned lb[F16_PX], sh[F16_PX], soff[F16_PX];
#pragma unroll
    for (int p = 0; p < F16_PX; ++p) {
        lb[p] = ring_lds + 4u * (unsigned)(px[p].lbase >> 1);
        sh[p] = (unsigned)(px[p].lbase & 1) * 16u;
        soff[p] = px[p].valid ? 2u * px[p].pix : 0x80000000u;   // (an invalid pixel's store is dropped by the range check)
    }
    const int last = c_end - 1;
    const __half* pdma = img + (int64_t)c_begin * cs;
    __half* pout = out + (int64_t)c_begin * cs;
    auto issue = [&](int slot) {
        const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)pdma, 0, plane_bytes, 0x00020000);
        unsigned* l = ring + slot * NP + wave_first;
#pragma unroll
        for (int k = 0; k < K; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (f16_lptr_t)(l + k * F16_THREADS), 4, goff[k], 0, 0, 0);
        pdma += cs;
    };
#define F16_READ2(dst, addr) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(dst) : "v"(addr))
#define F16_READ1(dst, addr) asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(dst) : "v"(addr))
    // The pipeline is skewed by one pixel across the barrier (filterinterp_lds.hip): in channel c a wave issues pixel 0's
    // reads, does the arithmetic of pixel 1 of channel c - 1 (taps read before the barrier, waiting in registers), issues
    // pixel 1's reads, does pixel 0, and waits for pixel 1's taps -- every LDS read is in flight under the wave's own arithmetic.
    v2u a01[2][4];          // dwords 0, 1 of the four tap rows, two pixels
    unsigned a2[2][4];      // dword 2
    auto reads = [&](int p, unsigned so) {
        unsigned a = lb[p] + so;
#pragma unroll
        for (int r = 0; r < 4; ++r) { F16_READ2(a01[p][r], a); F16_READ1(a2[p][r], a); a += pitch4; }
    };
    auto pixel = [&](int p, const __half* plane_ptr) {
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // ({d1,d0} >> sh) and ({d2,d1} >> sh): the four taps as two packed pairs
            const unsigned e0 = __builtin_amdgcn_alignbit(a01[p][r].y, a01[p][r].x, sh[p]);
            const unsigned e1 = __builtin_amdgcn_alignbit(a2[p][r], a01[p][r].y, sh[p]);
            acc = fmaf(f16_lo(e0), px[p].g[r * 4 + 0], acc);
            acc = fmaf(f16_hi(e0), px[p].g[r * 4 + 1], acc);
            acc = fmaf(f16_lo(e1), px[p].g[r * 4 + 2], acc);
            acc = fmaf(f16_hi(e1), px[p].g[r * 4 + 3], acc);
        }
        const auto oplane = __builtin_amdgcn_make_buffer_rsrc((void*)plane_ptr, 0, plane_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b16(__half_as_ushort(f16_store_value(acc)), oplane, soff[p], 0, 0);
    };
    auto compute = [&](int slot, bool first) {
        const unsigned so = (unsigned)(slot * (NP * 4));
        reads(0, so);
        if (!first) pixel(1, pout - cs);                    // pixel 1 of the previous channel
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");  // (at most 15 LDS reads outstanding)
        reads(1, so);
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a01[0][0]), "+v"(a01[0][1]), "+v"(a01[0][2]), "+v"(a01[0][3]),
                                               "+v"(a2[0][0]), "+v"(a2[0][1]), "+v"(a2[0][2]), "+v"(a2[0][3]));
        pixel(0, pout);
        // pixel 1's taps are in registers before the barrier: the slot may be overwritten after it
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a01[1][0]), "+v"(a01[1][1]), "+v"(a01[1][2]), "+v"(a01[1][3]),
                                               "+v"(a2[1][0]), "+v"(a2[1][1]), "+v"(a2[1][2]), "+v"(a2[1][3]));
        pout += cs;
    };
#undef F16_READ2
#undef F16_READ1
    const int n0 = min(D, c_end - c_begin);
    for (int j = 0; j < n0; ++j) issue(j);
    f16_wait_windows<K>(n0 - 1);                                // the first window has landed ...
    __builtin_amdgcn_s_barrier();                               // ... in every wave
    int c = c_begin, slot = 0;
    for (; c + D <= last; ++c) {                                // steady state: window c + D exists
        issue(slot == 0 ? R - 1 : slot - 1);                    // into the slot every wave finished reading before the last barrier
        compute(slot, c == c_begin);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * K) : "memory");      // all but the D - 1 youngest windows: c + 1 has landed
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    for (; c <= last; ++c) {                                    // the last D channels: nothing left to stage
        compute(slot, c == c_begin);
        if (c < last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    pixel(1, pout - cs);                                        // pixel 1 of the last channel
#pragma unroll
    for (int p = 0; p < F16_PX; ++p)
        if (px[p].inimg && !px[p].valid)
            for (int cc = c_begin; cc < c_end; ++cc) out[(int64_t)cc * cs + px[p].pix] = img[(int64_t)cc * cs + px[p].pix];
}

__global__ __launch_bounds__(F16_THREADS, 4) void fi_forward_ori_lds_f16(
    const __half* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    __half* __restrict__ out, int channel, int h, int w,
    vfi_strides s1, vfi_strides s2, vfi_strides s3,
    int tiles_x, int tiles_y, int ntiles, int ch_per_group) {
    __shared__ unsigned lds[F16_HDR + F16_RING_DWORDS];
    int* box = reinterpret_cast<int*>(lds);

    // four horizontally consecutive tiles on one XCD (workgroups are dealt round-robin to the 8 XCDs): the 128-byte lines
    // a window shares with its left and right neighbours are fetched into one L2 (filterinterp_lds.hip)
    const int xs = blockIdx.x % 8, kx = blockIdx.x / 8;
    const int tile = ((kx / 4) * 8 + xs) * 4 + (kx % 4);
    if (tile >= ntiles) return;
    const int b = tile / (tiles_x * tiles_y);
    const int trem = tile - b * (tiles_x * tiles_y);
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int c_begin = blockIdx.y * ch_per_group;
    const int c_end = min(channel, c_begin + ch_per_group);

    const int tid = threadIdx.x;
    const int x = txi * F16_TW + (tid & (F16_TW - 1));
    const int y0 = tyi * F16_TH + (tid >> 6);

    F16Pixel px[F16_PX];
    int L[F16_PX], Lc[F16_PX], T[F16_PX];
    float alpha[F16_PX], beta[F16_PX];
    int bx_lo = INT_MAX, by_lo = INT_MAX, bx_hi = INT_MIN, by_hi = INT_MIN;
#pragma unroll
    for (int p = 0; p < F16_PX; ++p) {
        const int y = y0 + p * F16_PASS_ROWS;
        px[p].inimg = x < w && y < h;
        px[p].pix = (unsigned)(y * (int)s1.h + x);
        float fx = 0.0f, fy = 0.0f;
        if (px[p].inimg) {
            const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
            fx = flow[0];
            fy = flow[s2.c];
        }
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        px[p].valid = px[p].inimg && fi_valid(fx, fy, x2, y2, w, h);
        const int ix = px[p].valid ? (int)x2 : 1, iy = px[p].valid ? (int)y2 : 0;
        L[p] = ix - 1;
        T[p] = iy - 1;
        Lc[p] = clampi(L[p], 0, w - 4);                     // the window moved inside the image (w >= 4)
        alpha[p] = x2 - (float)ix;
        beta[p] = y2 - (float)iy;
        if (px[p].valid) {
            bx_lo = min(bx_lo, Lc[p]); by_lo = min(by_lo, T[p]);
            bx_hi = max(bx_hi, Lc[p] + 3); by_hi = max(by_hi, T[p] + 3);
        }
    }

    if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
    __syncthreads();
    {
        const int x0 = wave_min_i32(bx_lo), y0w = wave_min_i32(by_lo);
        const int x1 = wave_max_i32(bx_hi), y1 = wave_max_i32(by_hi);
        if ((tid & 63) == 0 && x0 != INT_MAX) {
            atomicMin(&box[0], x0); atomicMin(&box[1], y0w);
            atomicMax(&box[2], x1); atomicMax(&box[3], y1);
        }
    }
    __syncthreads();
    const bool any_valid = box[0] != INT_MAX;
    const int bx0 = any_valid ? box[0] & ~1 : 0, by0 = box[1];          // even: dword-aligned rows
    const int bw = any_valid ? (box[2] - bx0 + 2) >> 1 : 0;              // dwords per row
    const int bh = any_valid ? box[3] - by0 + 1 : 0;
    const int pitch = fi_pitch_for(bw);
    const int n = pitch * bh;

    // ---- folded tap weights: filter tap x bilinear quadrant weight, clamped columns merged
#pragma unroll
    for (int p = 0; p < F16_PX; ++p) {
        px[p].lbase = px[p].valid ? (T[p] - by0) * (2 * pitch) + (Lc[p] - bx0) : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) px[p].g[k] = 0.0f;
        if (px[p].valid) {
            const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)(y0 + p * F16_PASS_ROWS) * s3.h + x;
            const float wx[2] = { 1.0f - alpha[p], alpha[p] }, wy[2] = { 1.0f - beta[p], beta[p] };
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float wgt = fpx[(int64_t)(r * 4 + k) * s3.c] * (wx[k >> 1] * wy[r >> 1]);
                    const int kk = clampi(L[p] + k, 0, w - 1) - Lc[p];          // 0..3
#pragma unroll
                    for (int q = 0; q < 4; ++q) px[p].g[r * 4 + q] += (kk == q) ? wgt : 0.0f;
                }
        }
    }

    const __half* img = in1 + (int64_t)b * s1.b;
    __half* dst = out + (int64_t)b * s1.b;
    const int kmax = (n + F16_THREADS - 1) / F16_THREADS;
    if (kmax > F16_KTOP) {
        // window too large for LDS: the same dot product gathered from global memory
#pragma unroll
        for (int p = 0; p < F16_PX; ++p) {
            if (px[p].valid) {
                for (int c = c_begin; c < c_end; ++c) {
                    const __half* plane = img + (int64_t)c * s1.c;
                    float acc = 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const __half* row = plane + (int64_t)clampi(T[p] + r, 0, h - 1) * s1.h + Lc[p];
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc = fmaf(__half2float(row[k]), px[p].g[r * 4 + k], acc);
                    }
                    dst[(int64_t)c * s1.c + px[p].pix] = f16_store_value(acc);
                }
            } else if (px[p].inimg) {
                for (int c = c_begin; c < c_end; ++c) dst[(int64_t)c * s1.c + px[p].pix] = img[(int64_t)c * s1.c + px[p].pix];
            }
        }
        return;
    }

    const F16Window win{bx0, by0, bw, bh, pitch, h, w, (int)s1.h};
    unsigned* ring = lds + F16_HDR;
#define F16_RUN(K) f16_run_channels<K>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring)
    if (kmax <= 2) F16_RUN(2);
    else if (kmax == 3) F16_RUN(3);
    else if (kmax == 4) F16_RUN(4);
    else if (kmax <= 6) F16_RUN(6);
    else F16_RUN(8);
#undef F16_RUN
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_filterinterp_forward_ori_f16_direct(const void* input1, const float* input2, const float* input3,
                                                        void* output, int batch, int channel, int h, int w,
                                                        int filter_channels, vfi_strides s1, vfi_strides s2,
                                                        vfi_strides s3, vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_channels <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !output) return VFI_ERR_SHAPE;
    const int fs = (int)sqrtf((float)filter_channels);
    hipLaunchKernelGGL(fi_forward_ori_direct_f16, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       (const __half*)input1, input2, input3, (__half*)output, channel, h, w, fs, s1, s2, s3);
    return launch_status();
}

extern "C" int vfi_filterinterp_forward_ori_f16(const void* input1, const float* input2, const float* input3,
                                                 void* output, int batch, int channel, int h, int w,
                                                 int filter_channels, vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                                 vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_channels <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !input3 || !output) return VFI_ERR_SHAPE;
    // the staged kernel moves dwords: rows, planes and the base address must be 4-byte aligned
    const bool staged = (int)sqrtf((float)filter_channels) == 4 && filter_channels == 16 && w >= 4 &&
                        ((s1.h | s1.c | s1.b) & 1) == 0 && (reinterpret_cast<uintptr_t>(input1) & 3) == 0 &&
                        (int64_t)h * s1.h * 2 < INT_MAX;
    const int tiles_x = (w + F16_TW - 1) / F16_TW, tiles_y = (h + F16_TH - 1) / F16_TH;
    const int64_t nt = (int64_t)tiles_x * tiles_y * batch;
    if (!staged || nt > (1 << 28))
        return vfi_filterinterp_forward_ori_f16_direct(input1, input2, input3, output, batch, channel, h, w,
                                                       filter_channels, s1, s2, s3, stream);
    const int ntiles = (int)nt;
    // split the channel range over blockIdx.y when that shortens the tail (as the fp32 kernel)
    const int best_groups = fi_channel_groups(ntiles, channel, 4.3);
    const int ch_per_group = (channel + best_groups - 1) / best_groups;
    const int groups = (channel + ch_per_group - 1) / ch_per_group;
    const int grid_x = ((ntiles + 31) / 32) * 32;                   // whole groups of 8 XCDs x 4 tiles
    hipLaunchKernelGGL(fi_forward_ori_lds_f16, dim3((unsigned)grid_x, (unsigned)groups, 1), dim3(F16_THREADS, 1, 1), 0,
                       (hipStream_t)stream, (const __half*)input1, input2, input3, (__half*)output, channel, h, w, s1, s2,
                       s3, tiles_x, tiles_y, ntiles, ch_per_group);
    return launch_status();
}
