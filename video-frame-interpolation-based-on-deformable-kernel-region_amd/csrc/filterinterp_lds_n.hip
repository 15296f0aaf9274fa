// filterinterp_lds_n.hip -- LDS-staged FilterInterpolation (_ori) forward for the filter sizes other than 4 that
// the reference's --filter_size offers (my_args.py: 2, 5, 6; 4 has its own tuned file, filterinterp_lds.hip).
//
// Semantics: filterinterpolation_cuda_kernel.cu:2692-2823; the arithmetic and its order are those of the direct
// kernel's general path (quadrants_generic, filterinterp_dev.h): quadrant by quadrant, rows outer / columns inner,
// every sum started at 0 and continued by fused multiply-adds.
//
// Same pipeline as the fs = 4 kernel: a workgroup owns a 64x8 tile (one pixel per thread: fs = 6 keeps 36 filter
// taps per pixel in registers), stages the bounding box of all its taps per channel by LDS-DMA into a ring of
// window slots (borders replicated while staging), counted vmcnt, one barrier per channel; a tile whose window does
// not fit gathers from global memory.  The general direct kernel re-reads the fs*fs filter taps and gathers fs*fs
// image values from global memory per pixel and channel.
#include "filterinterp_dev.h"

#include <limits.h>

namespace vfi {

#define FN_TW 64
#define FN_TH 8
#define FN_THREADS (FN_TW * FN_TH)
#define FN_HDR 16
#define FN_RING_FLOATS 15984
#define FN_RMAX 5
#define FN_KTOP 15

typedef __attribute__((address_space(3))) void* fn_lptr_t;

template <int K>
__device__ __forceinline__ void fn_wait_windows(int younger_groups) {
    switch (younger_groups) {
    case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
    case 2:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * K) : "memory"); break;
    }
}

struct FnWindow { int bx0, by0, bw, bh, pitch, h, w, hs; };

template <int FS>
struct FnPixel {
    bool valid, inimg;
    float alpha, beta;
    int lbase;              // LDS index of the pixel's window origin inside a staged window
    unsigned pix;
    float f[FS * FS];
};

// the four quadrant sums from a window whose rows are `pitch` floats apart (fetch(r, k) = tap of row r, column k)
template <int FS, typename F>
__device__ __forceinline__ float fn_value(const FnPixel<FS>& px, F&& fetch) {
    constexpr int HL = FS / 2;                              // rows T..iy and columns L..ix
    float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int quad = 0; quad < 4; ++quad) {
        const int r0 = (quad & 2) ? HL : 0, r1 = (quad & 2) ? FS : HL;
        const int k0 = (quad & 1) ? HL : 0, k1 = (quad & 1) ? FS : HL;
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < FS; ++r)
#pragma unroll
            for (int k = 0; k < FS; ++k)
                if (r >= r0 && r < r1 && k >= k0 && k < k1) acc = fmaf(fetch(r, k), px.f[r * FS + k], acc);
        q[quad] = acc;
    }
    return blend4(px.alpha, px.beta, q[0], q[1], q[2], q[3]);
}

template <int FS, int K>
__device__ __forceinline__ void fn_run_channels(const float* __restrict__ img, float* __restrict__ out, int64_t cs,
                                                int c_begin, int c_end, int tid, const FnWindow& win,
                                                const FnPixel<FS>& px, float* __restrict__ ring, int R) {
    static_assert(3 * K <= 63, "vmcnt is a 6-bit counter");
    unsigned goff[K];                                       // staging exactly as fi_run_channels (filterinterp_lds.hip)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + k * FN_THREADS;
        const int r = e / win.pitch;
        const int col = e - r * win.pitch;
        const unsigned off = 4u * (unsigned)(clampi(win.by0 + r, 0, win.h - 1) * win.hs + clampi(win.bx0 + col, 0, win.w - 1));
        goff[k] = (col < win.bw && r < win.bh) ? off : 0x80000000u;
    }
    const int plane_bytes = 4 * ((win.h - 1) * win.hs + win.w);
    constexpr int NP = K * FN_THREADS;
    const int D = R - 1;
    auto issue = [&](int c, int slot) {
        const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)c * cs), 0, plane_bytes, 0x00020000);
        float* l = ring + slot * NP + tid;
#pragma unroll
        for (int k = 0; k < K; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (fn_lptr_t)(l + k * FN_THREADS), 4, goff[k], 0, 0, 0);
    };
    auto compute = [&](int c, int slot) {
        if (!px.valid) return;
        const float* t = ring + slot * NP + px.lbase;
        const int pitch = win.pitch;
        out[(int64_t)c * cs + px.pix] = fn_value<FS>(px, [&](int r, int k) { return t[r * pitch + k]; });
    };
    if (c_begin >= c_end) return;
    const int last = c_end - 1;
    for (int j = 0; j < D; ++j)
        if (c_begin + j <= last) issue(c_begin + j, j);
    fn_wait_windows<K>(min(c_begin + D - 1, last) - c_begin);
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int c = c_begin; c <= last; ++c) {
        if (c + D <= last) issue(c + D, slot == 0 ? R - 1 : slot - 1);      // the slot read last iteration is free
        compute(c, slot);
        if (c < last) fn_wait_windows<K>(min(c + D, last) - (c + 1));
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    if (px.inimg && !px.valid)                               // copy-through (:2814-2818)
        for (int c = c_begin; c < c_end; ++c) out[(int64_t)c * cs + px.pix] = img[(int64_t)c * cs + px.pix];
}

template <int FS>
__global__ __launch_bounds__(FN_THREADS, 4) void fi_forward_ori_lds_n(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    float* __restrict__ out, int channel, int h, int w,
    vfi_strides s1, vfi_strides s2, vfi_strides s3,
    int tiles_x, int tiles_y, int ntiles, int ch_per_group) {
    __shared__ float lds[FN_HDR + FN_RING_FLOATS];          // one array: header (bounding box) + window ring
    int* box = reinterpret_cast<int*>(lds);
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int b = tile / (tiles_x * tiles_y);
    const int trem = tile - b * (tiles_x * tiles_y);
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int c_begin = blockIdx.y * ch_per_group;
    const int c_end = min(channel, c_begin + ch_per_group);
    const int tid = threadIdx.x;
    const int x = txi * FN_TW + (tid & (FN_TW - 1));
    const int y = tyi * FN_TH + (tid >> 6);

    FnPixel<FS> px;
    px.inimg = x < w && y < h;
    px.pix = (unsigned)(y * (int)s1.h + x);
    float fx = 0.0f, fy = 0.0f;
    if (px.inimg) {
        const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        fx = flow[0];
        fy = flow[s2.c];
    }
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    px.valid = px.inimg && fi_valid(fx, fy, x2, y2, w, h);
    const int ix = px.valid ? (int)x2 : 0, iy = px.valid ? (int)y2 : 0;
    const int L = ix + 1 - FS / 2, T = iy + 1 - FS / 2;
    px.alpha = x2 - (float)ix;
    px.beta = y2 - (float)iy;

    if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
    __syncthreads();
    {
        const int x0 = wave_min_i32(px.valid ? L : INT_MAX), y0w = wave_min_i32(px.valid ? T : INT_MAX);
        const int x1 = wave_max_i32(px.valid ? L + FS - 1 : INT_MIN), y1 = wave_max_i32(px.valid ? T + FS - 1 : INT_MIN);
        if ((tid & 63) == 0 && x0 != INT_MAX) {
            atomicMin(&box[0], x0); atomicMin(&box[1], y0w);
            atomicMax(&box[2], x1); atomicMax(&box[3], y1);
        }
    }
    __syncthreads();
    const int bx0 = box[0], by0 = box[1];
    const bool any_valid = bx0 != INT_MAX;
    const int bw = any_valid ? box[2] - bx0 + 1 : 0;
    const int bh = any_valid ? box[3] - by0 + 1 : 0;
    const int pitch = (bw + 31) & ~31;
    const int n = pitch * bh;
    px.lbase = (T - by0) * pitch + (L - bx0);
    if (px.valid) {
        const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
#pragma unroll
        for (int k = 0; k < FS * FS; ++k) px.f[k] = fpx[(int64_t)k * s3.c];
    } else {
#pragma unroll
        for (int k = 0; k < FS * FS; ++k) px.f[k] = 0.0f;
    }

    const float* img = in1 + (int64_t)b * s1.b;
    float* dst = out + (int64_t)b * s1.b;
    const int hs = (int)s1.h;
    const int kmax = (n + FN_THREADS - 1) / FN_THREADS;
    if (kmax > FN_KTOP) {
        // window too large for LDS: gather from global memory (workgroup-uniform branch), same arithmetic
        if (px.valid) {
            const int plane_bytes = 4 * ((h - 1) * hs + w);
            for (int c = c_begin; c < c_end; ++c) {
                const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)c * s1.c), 0, plane_bytes, 0x00020000);
                dst[(int64_t)c * s1.c + px.pix] = fn_value<FS>(px, [&](int r, int k) {
                    int ty = T, tx = L;
                    asm volatile("" : "+v"(ty), "+v"(tx));  // tap addresses re-derived per channel, not hoisted (registers)
                    const unsigned o = 4u * (unsigned)(clampi(ty + r, 0, h - 1) * hs + clampi(tx + k, 0, w - 1));
                    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, o, 0, 0));
                });
            }
        } else if (px.inimg) {
            for (int c = c_begin; c < c_end; ++c) dst[(int64_t)c * s1.c + px.pix] = img[(int64_t)c * s1.c + px.pix];
        }
        return;
    }
    const FnWindow win{bx0, by0, bw, bh, pitch, h, w, hs};
    float* ring = lds + FN_HDR;
#define FN_RUN(K) fn_run_channels<FS, K>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring, \
                                         min(FN_RMAX, FN_RING_FLOATS / ((K) * FN_THREADS)))
    if (kmax <= 3) FN_RUN(3);
    else if (kmax == 4) FN_RUN(4);
    else if (kmax == 5) FN_RUN(5);
    else if (kmax == 6) FN_RUN(6);
    else if (kmax <= 8) FN_RUN(8);
    else if (kmax <= 10) FN_RUN(10);
    else if (kmax <= 12) FN_RUN(12);
    else FN_RUN(15);
#undef FN_RUN
}

}  // namespace vfi

using namespace vfi;

// returns -1 when this path does not apply (the caller uses the direct kernel)
extern "C" int vfi_filterinterp_forward_ori_lds_n(const float* input1, const float* input2, const float* input3,
                                                   float* output, int batch, int channel, int h, int w, int fs,
                                                   vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                                   vfi_stream_t stream) {
    if (!(fs == 2 || fs == 5 || fs == 6)) return -1;
    if ((int64_t)h * s1.h * 4 > INT_MAX) return -1;          // byte offsets inside a plane are 32-bit
    const int tiles_x = (w + FN_TW - 1) / FN_TW, tiles_y = (h + FN_TH - 1) / FN_TH;
    const int64_t nt = (int64_t)tiles_x * tiles_y * batch;
    if (nt > INT_MAX) return -1;
    const int ntiles = (int)nt;
    // (the prologue reads flow + fs x fs filter planes: ~4.3 channels' worth at fs = 4)
    const int best_groups = fi_channel_groups(ntiles, channel, 4.3 * (2 + fs * fs) / 18.0);
    const int ch_per_group = (channel + best_groups - 1) / best_groups;
    const int groups = (channel + ch_per_group - 1) / ch_per_group;
    const dim3 grid((unsigned)ntiles, (unsigned)groups, 1), block(FN_THREADS, 1, 1);
    hipStream_t st = (hipStream_t)stream;
    if (fs == 2)
        hipLaunchKernelGGL(fi_forward_ori_lds_n<2>, grid, block, 0, st, input1, input2, input3, output, channel, h, w,
                           s1, s2, s3, tiles_x, tiles_y, ntiles, ch_per_group);
    else if (fs == 5)
        hipLaunchKernelGGL(fi_forward_ori_lds_n<5>, grid, block, 0, st, input1, input2, input3, output, channel, h, w,
                           s1, s2, s3, tiles_x, tiles_y, ntiles, ch_per_group);
    else
        hipLaunchKernelGGL(fi_forward_ori_lds_n<6>, grid, block, 0, st, input1, input2, input3, output, channel, h, w,
                           s1, s2, s3, tiles_x, tiles_y, ntiles, ch_per_group);
    return launch_status();
}
