// filterinterp_defor_lds.hip -- LDS-staged forward of the deformable FilterInterpolation variants, fs == 4 and fs == 6
// (the two sizes the reference's 4-input kernel has bodies for, filterinterpolation_cuda_kernel.cu:68).
//
// Semantics: filterinterpolation_cuda_kernel.cu:29-426 (4-input forward, VARIANT 0), :1353-1496
// (deforconv = the paper's deformable kernel region, VARIANT 1), :2070-2191 (nofilterwithdeforconv,
// VARIANT 2); the same arithmetic, in the same order, as fi_forward_defor / fi_forward_defor4 in
// filterinterp.hip -- only where the image values come from differs.
//
// Why: every one of the 16 taps of a pixel is a bilinear sample at (tap + learned offset): 64 image values
// per pixel and channel, at addresses that jitter from lane to lane.  Gathered from global memory that is 64
// vector loads per pixel and channel, each touching several cache lines: 0.13 ms per channel at 1080p however
// little arithmetic surrounds them (measured: hoisting all tap geometry out of the channel loop changed
// nothing).  Here a workgroup owns a 64x8 tile of output pixels, takes the bounding box of all corners of all
// taps of the tile and stages that window of each image plane into LDS by LDS-DMA, exactly as the _ori kernel
// does (filterinterp_lds.hip: ring of window slots, counted vmcnt, one barrier per channel, borders
// replicated while staging so that a corner index is never clamped again).  A tap is then two ds_read2_b32.
// What stays in registers for all channels: per tap the LDS index of its top-left corner and the two
// bilinear fractions, the filter weight, and its quadrant.
//
// A tile whose window does not fit the LDS budget (large learned offsets) gathers from global memory instead,
// decided per workgroup; other filter sizes use fi_forward_defor.  fs == 6 keeps 36 taps x (corner index, two fractions,
// weight) = 144 registers per pixel: one workgroup per CU (at 256 registers the region variant spills inside its pipeline).
#include "filterinterp_dev.h"

#include <limits.h>

namespace vfi {

#define DF_TW 64
#define DF_TH 4
#define DF_THREADS (DF_TW * DF_TH)                  // 256: one pixel per thread
#define DF_HDR 16
#define DF_RING_FLOATS 13040
#define DF_RMAX 5
#define DF_KTOP 15

typedef __attribute__((address_space(3))) void* df_lptr_t;

template <int K>
__device__ __forceinline__ void df_wait_windows(int younger_groups) {
    switch (younger_groups) {
    case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); break;
    case 2:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * K) : "memory"); break;
    }
}

struct DfWindow { int bx0, by0, bw, bh, pitch, h, w, hs; };

// One pixel's state for the channel loop.  lb[k]: float index, inside a staged window, of tap k's top-left
// corner; phy / phx: the fractions of defor_tap; qx / qy bit k: the tap lies right of / below the sampling
// position (VARIANT 1, 2).
template <int FS> struct DfPixel {
    bool valid, inimg;
    float alpha, beta;
    unsigned pix;
    unsigned long long qx, qy;
    int lb[FS * FS];
    float phy[FS * FS], phx[FS * FS], wgt[FS * FS];
};

// one tap: bilinear sample from its four corners (defor_tap's arithmetic, filterinterp.hip), then the
// quadrant sum it belongs to
template <int VARIANT, int FS, int KTAP>
__device__ __forceinline__ void df_tap(const DfPixel<FS>& px, float a, float b, float c, float d, float (&q)[4]) {
    float phiY = px.phy[KTAP], phiX = px.phx[KTAP];
    // keep the four corner weights out of the registers: left alone, the compiler hoists all 64 of them (and
    // the 16 second-row addresses) out of the channel loop and spills hundreds of registers
    asm volatile("" : "+v"(phiY), "+v"(phiX));
    float v = ((1.0f - phiX) * (1.0f - phiY)) * a;
    v = fmaf(phiX * (1.0f - phiY), b, v);
    v = fmaf((1.0f - phiX) * phiY, c, v);
    v = fmaf(phiY * phiX, d, v);
    if constexpr (VARIANT == VFI_DEFOR_OFFSET) {
        constexpr int quad = ((KTAP / FS) >= FS / 2 ? 2 : 0) + ((KTAP % FS) >= FS / 2 ? 1 : 0);     // by integer index
        q[quad] = fmaf(v, px.wgt[KTAP], q[quad]);
    } else {
        // by displaced position.  The quadrant number is re-extracted from the packed bits per channel: hoisted,
        // the 64 loop-invariant lane masks (16 taps x 4 quadrants) overflow the scalar registers and come back as
        // hundreds of v_readlane / v_writelane
        unsigned code = (unsigned)((px.qx >> KTAP) & 1ull) | ((unsigned)((px.qy >> KTAP) & 1ull) << 1);
        asm volatile("" : "+v"(code));
#pragma unroll
        for (int quad = 0; quad < 4; ++quad) {
            const bool mine = code == (unsigned)quad;
            const float upd = (VARIANT == VFI_DEFOR_NOFILTER) ? q[quad] + v : fmaf(v, px.wgt[KTAP], q[quad]);
            q[quad] = mine ? upd : q[quad];
        }
    }
}

// a row of FS taps: its 4 FS corner values are fetched (F: tap index -> the four corner values) before they are used
template <int VARIANT, int FS, int J, int I, typename V>
__device__ __forceinline__ void df_tap_row_taps(const DfPixel<FS>& px, const V& v, float (&q)[4]) {
    if constexpr (I < FS) {
        df_tap<VARIANT, FS, J * FS + I>(px, v[I][0], v[I][1], v[I][2], v[I][3], q);
        df_tap_row_taps<VARIANT, FS, J, I + 1>(px, v, q);
    }
}
template <int VARIANT, int FS, int J, typename F>
__device__ __forceinline__ void df_tap_rows(const DfPixel<FS>& px, F&& fetch, float (&q)[4]) {
    if constexpr (J < FS) {
        float v[FS][4];
#pragma unroll
        for (int i = 0; i < FS; ++i) fetch(J * FS + i, v[i]);
        df_tap_row_taps<VARIANT, FS, J, 0>(px, v, q);
        df_tap_rows<VARIANT, FS, J + 1>(px, fetch, q);
    }
}

template <int VARIANT, int FS, typename F>
__device__ __forceinline__ float df_value(const DfPixel<FS>& px, F&& fetch) {
    float q[4] = {0.0f, 0.0f, 0.0f, 0.0f};                  // TL, TR, BL, BR
    df_tap_rows<VARIANT, FS, 0>(px, fetch, q);
    return blend4(px.alpha, px.beta, q[0], q[1], q[2], q[3]);
}

template <int VARIANT, int FS, int K>
__device__ __forceinline__ void df_run_channels(const float* __restrict__ img, float* __restrict__ out, int64_t cs,
                                                int c_begin, int c_end, int tid, const DfWindow& win,
                                                const DfPixel<FS>& px, float* __restrict__ ring, int R) {
    static_assert(3 * K <= 63, "vmcnt is a 6-bit counter");
    // staging exactly as fi_run_channels (filterinterp_lds.hip): element e = tid + k * threads of the window,
    // row pitch a multiple of the 32 banks, pad elements get an out-of-range offset (no memory traffic)
    unsigned goff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + k * DF_THREADS;
        const int r = e / win.pitch;
        const int col = e - r * win.pitch;
        const unsigned off = 4u * (unsigned)(clampi(win.by0 + r, 0, win.h - 1) * win.hs + clampi(win.bx0 + col, 0, win.w - 1));
        goff[k] = (col < win.bw && r < win.bh) ? off : 0x80000000u;
    }
    const int plane_bytes = 4 * ((win.h - 1) * win.hs + win.w);
    constexpr int NP = K * DF_THREADS;
    const int D = R - 1;
    auto issue = [&](int c, int slot) {
        const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)c * cs), 0, plane_bytes, 0x00020000);
        float* l = ring + slot * NP + tid;
#pragma unroll
        for (int k = 0; k < K; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (df_lptr_t)(l + k * DF_THREADS), 4, goff[k], 0, 0, 0);
    };
    auto compute = [&](int c, int slot) {
        if (!px.valid) return;
        const float* base = ring + slot * NP;
        const int pitch = win.pitch;
        out[(int64_t)c * cs + px.pix] = df_value<VARIANT, FS>(px, [&](int k, float (&v)[4]) {
            int o = px.lb[k];
            asm volatile("" : "+v"(o));                     // (second-row address re-derived per channel, see df_tap)
            const float* t = base + o;
            v[0] = t[0]; v[1] = t[1];                       // ds_read2_b32 at (0, 1), twice
            v[2] = t[pitch]; v[3] = t[pitch + 1];
        });
    };
    if (c_begin >= c_end) return;
    const int last = c_end - 1;
    for (int j = 0; j < D; ++j)
        if (c_begin + j <= last) issue(c_begin + j, j);
    df_wait_windows<K>(min(c_begin + D - 1, last) - c_begin);
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int c = c_begin; c <= last; ++c) {
        if (c + D <= last) issue(c + D, slot == 0 ? R - 1 : slot - 1);      // the slot read last iteration is free
        compute(c, slot);
        if (c < last) df_wait_windows<K>(min(c + D, last) - (c + 1));
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    if (px.inimg && !px.valid)                               // copy-through of the out-of-range pixels
        for (int c = c_begin; c < c_end; ++c) out[(int64_t)c * cs + px.pix] = img[(int64_t)c * cs + px.pix];
}

template <int VARIANT, int FS>
__global__ __launch_bounds__(DF_THREADS, FS == 4 ? 3 : 1) void fi_forward_defor_lds(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    const float* __restrict__ in4, float* __restrict__ out, int channel, int h, int w,
    vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4,
    int tiles_x, int tiles_y, int ntiles, int ch_per_group, unsigned filt_bytes, unsigned off_bytes) {
    __shared__ float lds[DF_HDR + DF_RING_FLOATS];          // one array: header (bounding box) + window ring
    int* box = reinterpret_cast<int*>(lds);
    const int tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int b = tile / (tiles_x * tiles_y);
    const int trem = tile - b * (tiles_x * tiles_y);
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int c_begin = blockIdx.y * ch_per_group;
    const int c_end = min(channel, c_begin + ch_per_group);
    const int tid = threadIdx.x;
    const int x = txi * DF_TW + (tid & (DF_TW - 1));
    const int y = tyi * DF_TH + (tid >> 6);

    constexpr int NT = FS * FS;                             // taps
    DfPixel<FS> px;
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
    px.qx = 0ull; px.qy = 0ull;

    // ---- the displaced taps of this pixel
    // top-left corner of each tap (frame coordinates, from -1), row and column packed into one register: with 36 taps the
    // two arrays beside the 3 x 36 per-tap constants do not fit the register file
    constexpr bool PACKED = FS != 4;                        // (fs == 4 keeps them apart: packed, that kernel spills inside its pipeline)
    unsigned tc[NT];
    int tt4[PACKED ? 1 : NT], tl4[PACKED ? 1 : NT];
    auto corner_y = [&](int k) { if constexpr (PACKED) return (int)(tc[k] >> 16) - 1; else return tt4[k]; };
    auto corner_x = [&](int k) { if constexpr (PACKED) return (int)(tc[k] & 0xffffu) - 1; else return tl4[k]; };
    int bx_lo = INT_MAX, by_lo = INT_MAX, bx_hi = INT_MIN, by_hi = INT_MIN;
    if (px.valid) {
        // 32 (48) loads first.  Buffer form: a wave-uniform descriptor of this batch item's filter / offset tensor,
        // the channel as scalar offset, the pixel as the one vector offset -- no 64-bit address pair per load
        // (48 of them in flight cost 96 registers and spilled)
        const auto frs = __builtin_amdgcn_make_buffer_rsrc((void*)(in3 + (int64_t)b * s3.b), 0, (int)filt_bytes, 0x00020000);
        const auto ors = (VARIANT == VFI_DEFOR_NOFILTER)
                             ? frs
                             : __builtin_amdgcn_make_buffer_rsrc((void*)(in4 + (int64_t)b * s4.b), 0, (int)off_bytes, 0x00020000);
        const unsigned fvo = 4u * (unsigned)(y * (int)s3.h + x);
        const unsigned ovo = (VARIANT == VFI_DEFOR_NOFILTER) ? fvo : 4u * (unsigned)(y * (int)s4.h + x);
        const unsigned ocs4 = 4u * (unsigned)((VARIANT == VFI_DEFOR_NOFILTER) ? s3.c : s4.c), fcs4 = 4u * (unsigned)s3.c;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            px.phy[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, ovo, k * ocs4, 0));
            px.phx[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ors, ovo, (NT + k) * ocs4, 0));
            px.wgt[k] = (VARIANT == VFI_DEFOR_NOFILTER)
                            ? 1.0f
                            : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frs, fvo, k * fcs4, 0));
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int cj = clampi(T + k / FS, 0, h - 1), ci = clampi(L + k % FS, 0, w - 1);
            const float fracY = (float)cj + px.phy[k];
            const float fracX = (float)ci + px.phx[k];
            const int Top = (int)fracY, Left = (int)fracX;
            px.phy[k] = fracY - (float)Top;
            px.phx[k] = fracX - (float)Left;
            if (fracX > x2) px.qx |= 1ull << k;
            if (fracY > y2) px.qy |= 1ull << k;
            // rows <= -1 all replicate row 0 and rows >= h - 1 row h - 1, so a corner pair starting at
            // clamp(Top, -1, h - 1) reads what clamp(Top), clamp(Top + 1) read: the window stays near the frame
            const int tt = clampi(Top, -1, h - 1), tl = clampi(Left, -1, w - 1);
            if constexpr (PACKED) tc[k] = ((unsigned)(tt + 1) << 16) | (unsigned)(tl + 1);
            else { tt4[k] = tt; tl4[k] = tl; }
            bx_lo = min(bx_lo, tl); by_lo = min(by_lo, tt);
            bx_hi = max(bx_hi, tl + 1); by_hi = max(by_hi, tt + 1);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            px.phy[k] = 0.0f; px.phx[k] = 0.0f; px.wgt[k] = 0.0f;
            if constexpr (PACKED) tc[k] = 0x00010001u; else { tt4[k] = 0; tl4[k] = 0; }
        }
    }

    // ---- bounding box of every corner of the tile
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
    const int bx0 = box[0], by0 = box[1];
    const bool any_valid = bx0 != INT_MAX;
    const int bw = any_valid ? box[2] - bx0 + 1 : 0;
    const int bh = any_valid ? box[3] - by0 + 1 : 0;
    const int pitch = (bw + 31) & ~31;
    const int64_t n64 = (int64_t)pitch * bh;
    const int kmax = (int)min((n64 + DF_THREADS - 1) / DF_THREADS, (int64_t)(DF_KTOP + 1));

    const float* img = in1 + (int64_t)b * s1.b;
    float* dst = out + (int64_t)b * s1.b;
    const int hs = (int)s1.h;
    if (kmax > DF_KTOP) {
        // window too large for LDS: gather from global memory (workgroup-uniform branch), same arithmetic
        if (px.valid) {
            const int plane_bytes = 4 * ((h - 1) * hs + w);
            for (int c = c_begin; c < c_end; ++c) {
                const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)c * s1.c), 0, plane_bytes, 0x00020000);
                dst[(int64_t)c * s1.c + px.pix] = df_value<VARIANT, FS>(px, [&](int k, float (&v)[4]) {
                    int ty = corner_y(k), tx = corner_x(k);
                    asm volatile("" : "+v"(ty), "+v"(tx));  // corner addresses re-derived per channel, not hoisted (registers)
                    const int r0 = clampi(ty, 0, h - 1) * hs, r1 = clampi(ty + 1, 0, h - 1) * hs;
                    const int q0 = clampi(tx, 0, w - 1), q1 = clampi(tx + 1, 0, w - 1);
                    v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, 4u * (unsigned)(r0 + q0), 0, 0));
                    v[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, 4u * (unsigned)(r0 + q1), 0, 0));
                    v[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, 4u * (unsigned)(r1 + q0), 0, 0));
                    v[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane, 4u * (unsigned)(r1 + q1), 0, 0));
                });
            }
        } else if (px.inimg) {
            for (int c = c_begin; c < c_end; ++c) dst[(int64_t)c * s1.c + px.pix] = img[(int64_t)c * s1.c + px.pix];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) px.lb[k] = (corner_y(k) - by0) * pitch + (corner_x(k) - bx0);

    const DfWindow win{bx0, by0, bw, bh, pitch, h, w, hs};
    float* ring = lds + DF_HDR;
#define DF_RUN(K) df_run_channels<VARIANT, FS, K>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring, \
                                              min(DF_RMAX, DF_RING_FLOATS / ((K) * DF_THREADS)))
    if (kmax <= 3) DF_RUN(3);
    else if (kmax == 4) DF_RUN(4);
    else if (kmax == 5) DF_RUN(5);
    else if (kmax == 6) DF_RUN(6);
    else if (kmax <= 8) DF_RUN(8);
    else if (kmax <= 10) DF_RUN(10);
    else if (kmax <= 12) DF_RUN(12);
    else DF_RUN(15);
#undef DF_RUN
}

}  // namespace vfi

using namespace vfi;

// returns -1 when this path does not apply (the caller uses the direct kernels)
extern "C" int vfi_filterinterp_forward_defor_lds(int variant, const float* input1, const float* input2,
                                                   const float* input3, const float* input4, float* output,
                                                   int batch, int channel, int h, int w, int filter_size,
                                                   vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4,
                                                   vfi_stream_t stream) {
    if (filter_size != 4 && filter_size != 6) return -1;
    if (h > 65534 || w > 65534) return -1;                   // (tap corners travel as two 16-bit halves)
    const int nt = filter_size * filter_size;
    if ((int64_t)h * s1.h * 4 > INT_MAX) return -1;          // byte offsets inside a plane are 32-bit
    // the filter / offset tensors of one batch item are addressed through 32-bit buffer offsets (signed descriptor size)
    const int nf = (variant == VFI_DEFOR_NOFILTER) ? 2 * nt : nt;
    const int64_t fb = 4 * ((int64_t)(nf - 1) * s3.c + (int64_t)(h - 1) * s3.h + w);
    const int64_t ob = (variant == VFI_DEFOR_NOFILTER) ? fb : 4 * ((int64_t)(2 * nt - 1) * s4.c + (int64_t)(h - 1) * s4.h + w);
    if (fb > INT_MAX || ob > INT_MAX || s3.c < 0 || s3.h < 0 || s4.c < 0 || s4.h < 0) return -1;
    const unsigned filt_bytes = (unsigned)fb, off_bytes = (unsigned)ob;
    const int tiles_x = (w + DF_TW - 1) / DF_TW, tiles_y = (h + DF_TH - 1) / DF_TH;
    const int64_t ntl = (int64_t)tiles_x * tiles_y * batch;
    if (ntl > INT_MAX) return -1;
    const int ntiles = (int)ntl;
    // (a channel of these kernels costs ~4x one of the _ori kernel, the prologue -- flow, offsets, filter: up to 200 B/pixel --
    //  ~3x: about 3 channels' worth)
    const int best_groups = fi_channel_groups(ntiles, channel, variant == VFI_DEFOR_NOFILTER ? 2.0 : 3.0);
    const int ch_per_group = (channel + best_groups - 1) / best_groups;
    const int groups = (channel + ch_per_group - 1) / ch_per_group;
    const dim3 grid((unsigned)ntiles, (unsigned)groups, 1), block(DF_THREADS, 1, 1);
    hipStream_t st = (hipStream_t)stream;
#define DF_LAUNCH(V, FSZ, IN4, S4, OB) hipLaunchKernelGGL((fi_forward_defor_lds<V, FSZ>), grid, block, 0, st, input1, input2, input3, IN4, \
                           output, channel, h, w, s1, s2, s3, S4, tiles_x, tiles_y, ntiles, ch_per_group, filt_bytes, OB)
    switch (variant) {
    case VFI_DEFOR_OFFSET:
        if (filter_size == 4) DF_LAUNCH(VFI_DEFOR_OFFSET, 4, input4, s4, off_bytes); else DF_LAUNCH(VFI_DEFOR_OFFSET, 6, input4, s4, off_bytes);
        break;
    case VFI_DEFOR_REGION:
        if (filter_size == 4) DF_LAUNCH(VFI_DEFOR_REGION, 4, input4, s4, off_bytes); else DF_LAUNCH(VFI_DEFOR_REGION, 6, input4, s4, off_bytes);
        break;
    case VFI_DEFOR_NOFILTER:
        if (filter_size == 4) DF_LAUNCH(VFI_DEFOR_NOFILTER, 4, input3, s3, filt_bytes); else DF_LAUNCH(VFI_DEFOR_NOFILTER, 6, input3, s3, filt_bytes);
        break;
    default:
        return -1;
    }
#undef DF_LAUNCH
    return launch_status();
}
