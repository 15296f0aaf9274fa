// filterinterp_lds.hip -- LDS-staged FilterInterpolation (_ori, fs == 4) forward.
//
// Semantics: filterinterpolation_cuda_kernel.cu:2692-2823 (same arithmetic as
// fi_forward_ori_direct; only where the image taps come from differs).
//
// Why: the op is a per-pixel 4x4 gather from the image at a flow-displaced
// position.  Done straight from global memory it costs 16 scattered loads per
// pixel and channel; for the 196-channel context tensors of DAIN_slowmotion
// that is the whole run time.  Here a workgroup owns a 64x16 tile of output
// pixels (one wave = one 64-pixel row: flow, the 16 filter planes and the output
// move as full 256-B rows), finds the bounding box of all its taps from the
// flow, and per channel stages exactly that window of the image plane into LDS
// with coalesced row reads (borders replicated while staging, so a pixel's 4x4
// window is always 4 contiguous floats x 4 rows in LDS).  The 16 taps then are
// LDS reads at immediate offsets.  Flow, blend weights, the 16 filter taps and
// the LDS row addresses stay in registers for all channels.  The window of
// channel c+1 is in flight (global -> registers) while channel c is computed
// (double-buffered LDS, one barrier per channel).
//
// A tile whose tap window does not fit the LDS budget (wildly divergent flow)
// gathers from global memory instead -- same results, decided per workgroup.
//
// Launch: 1-D grid; block b -> tile so that the blocks of one XCD (b % 8) own a
// contiguous band of tiles: neighbouring tiles re-read each other's halo rows
// from the same L2.  An optional split of the channel range over blockIdx.y
// shortens the tail when tiles * 1 does not fill the chip evenly.
#include "filterinterp_dev.h"

#include <limits.h>

namespace vfi {

#define FI_TW 64
#define FI_TH 16
#define FI_THREADS (FI_TW * FI_TH)
#define FI_CAP 8000                 // floats per LDS buffer (2 buffers = 64,000 B)
#define FI_NPT ((FI_CAP + FI_THREADS - 1) / FI_THREADS)
#define FI_XCDS 8

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(FI_THREADS) void fi_forward_ori_lds(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    float* __restrict__ out, int channel, int h, int w,
    vfi_strides s1, vfi_strides s2, vfi_strides s3,
    int tiles_x, int tiles_y, int ntiles, int per_xcd, int ch_per_group) {
    __shared__ float buf[2][FI_CAP];
    __shared__ int box[4];

    // ---- block -> tile (XCD-contiguous bands)
    const int bid = blockIdx.x;
    const int tile = (bid % FI_XCDS) * per_xcd + bid / FI_XCDS;
    if (tile >= ntiles) return;                             // whole workgroup leaves together
    const int b = tile / (tiles_x * tiles_y);
    const int trem = tile - b * (tiles_x * tiles_y);
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int c_begin = blockIdx.y * ch_per_group;
    const int c_end = min(channel, c_begin + ch_per_group);

    const int tid = threadIdx.x;
    const int x = txi * FI_TW + (tid & (FI_TW - 1));
    const int y = tyi * FI_TH + (tid >> 6);
    const bool inimg = x < w && y < h;

    // ---- this thread's pixel: flow, validity, window origin, blend weights
    float fx = 0.0f, fy = 0.0f;
    if (inimg) {
        const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        fx = flow[0];
        fy = flow[s2.c];
    }
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    const bool valid = inimg && fi_valid(fx, fy, x2, y2, w, h);
    const int ix = valid ? (int)x2 : 0, iy = valid ? (int)y2 : 0;
    const int L = ix - 1, T = iy - 1;                       // ix + 1 - fs/2, fs == 4
    const float alpha = x2 - (float)ix;
    const float beta = y2 - (float)iy;

    // ---- bounding box of every tap of the tile (unclamped window coordinates)
    if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
    __syncthreads();
    {
        const int x0 = wave_min(valid ? L : INT_MAX), y0 = wave_min(valid ? T : INT_MAX);
        const int x1 = wave_max(valid ? L + 3 : INT_MIN), y1 = wave_max(valid ? T + 3 : INT_MIN);
        if ((tid & 63) == 0 && x0 != INT_MAX) {
            atomicMin(&box[0], x0); atomicMin(&box[1], y0);
            atomicMax(&box[2], x1); atomicMax(&box[3], y1);
        }
    }
    __syncthreads();
    const int bx0 = box[0], by0 = box[1];
    const bool any_valid = bx0 != INT_MAX;
    const int bw = any_valid ? box[2] - bx0 + 1 : 0;
    const int bh = any_valid ? box[3] - by0 + 1 : 0;
    const int n = bw * bh;                                  // <= (w+2)*(h+2): fits int for any real frame

    // ---- the 16 filter taps of this pixel
    float f[16];
    if (valid) {
        const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = fpx[(int64_t)k * s3.c];
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) f[k] = 0.0f;
    }

    const float* img = in1 + (int64_t)b * s1.b;
    const int64_t pix = (int64_t)y * s1.h + x;
    float* dst = out + (int64_t)b * s1.b + pix;

    if (n > FI_CAP) {
        // window too large for LDS: gather from global memory (workgroup-uniform branch)
        if (valid) {
            fi4_channels_direct(img, dst, c_begin, c_end, s1.c, (int)s1.h, h, w, L, T, f, alpha, beta);
        } else if (inimg) {
            for (int c = c_begin; c < c_end; ++c) dst[(int64_t)c * s1.c] = img[(int64_t)c * s1.c + pix];
        }
        return;
    }

    // ---- staging plan: element e = tid + k*1024 of the window, row-major with row length bw
    int goff[FI_NPT];
#pragma unroll
    for (int k = 0; k < FI_NPT; ++k) {
        const int e = tid + k * FI_THREADS;
        const int r = (bw > 0) ? e / bw : 0;
        const int col = e - r * bw;
        goff[k] = clampi(by0 + r, 0, h - 1) * (int)s1.h + clampi(bx0 + col, 0, w - 1);
    }
    const int kmax = (n + FI_THREADS - 1) / FI_THREADS;     // workgroup-uniform trip count
    // LDS address of this pixel's window origin; rows are bw floats apart
    const int lbase = (T - by0) * bw + (L - bx0);

    float stage[FI_NPT];
    auto load_window = [&](int c) {
        const float* p = img + (int64_t)c * s1.c;
#pragma unroll
        for (int k = 0; k < FI_NPT; ++k)
            if (k < kmax && tid + k * FI_THREADS < n) stage[k] = p[goff[k]];
    };
    auto store_window = [&](int which) {
#pragma unroll
        for (int k = 0; k < FI_NPT; ++k)
            if (k < kmax && tid + k * FI_THREADS < n) buf[which][tid + k * FI_THREADS] = stage[k];
    };

    if (c_begin < c_end) {
        load_window(c_begin);
        store_window(0);
        if (c_begin + 1 < c_end) load_window(c_begin + 1);
    }
    __syncthreads();
    for (int c = c_begin; c < c_end; ++c) {
        const int cur = (c - c_begin) & 1;
        if (valid) {
            const float* t = &buf[cur][lbase];
            float v[16];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[r * 4 + k] = t[r * bw + k];
            dst[(int64_t)c * s1.c] = fi4_pixel(v, f, alpha, beta);
        } else if (inimg) {
            dst[(int64_t)c * s1.c] = img[(int64_t)c * s1.c + pix];     // copy-through (:2814-2818)
        }
        if (c + 1 < c_end) store_window(cur ^ 1);
        if (c + 2 < c_end) load_window(c + 2);
        __syncthreads();
    }
}

}  // namespace vfi

using namespace vfi;

static int fi_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    return cus;
}

// returns -1 when this path does not apply (caller uses the direct kernel)
extern "C" int vfi_filterinterp_forward_ori_lds(const float* input1, const float* input2, const float* input3,
                                                 float* output, int batch, int channel, int h, int w,
                                                 vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                                 vfi_stream_t stream) {
    // plane offsets inside the kernel are 32-bit
    if ((int64_t)h * s1.h > INT_MAX) return -1;
    const int tiles_x = (w + FI_TW - 1) / FI_TW, tiles_y = (h + FI_TH - 1) / FI_TH;
    const int64_t nt = (int64_t)tiles_x * tiles_y * batch;
    if (nt > (1 << 28)) return -1;
    const int ntiles = (int)nt;
    const int per_xcd = (ntiles + FI_XCDS - 1) / FI_XCDS;

    // split the channel range over blockIdx.y when that shortens the tail: two
    // 1024-thread workgroups per CU run at a time; every extra group re-reads the
    // flow + 16 filter planes (72 B/pixel) next to 8 B/pixel/channel of image traffic
    const int slots = fi_cu_count() * 2;
    int best_groups = 1;
    double best_cost = 0.0;
    for (int g = 1; g <= 8 && g <= channel; g *= 2) {
        const double wgs = (double)ntiles * g;
        const double tail = ceil(wgs / slots) * slots / wgs;                // >= 1
        const double bytes = (72.0 * g + 8.0 * channel) / (72.0 + 8.0 * channel);
        const double cost = tail * bytes;
        if (g == 1 || cost < best_cost) { best_cost = cost; best_groups = g; }
    }
    const int ch_per_group = (channel + best_groups - 1) / best_groups;
    const int groups = (channel + ch_per_group - 1) / ch_per_group;

    const dim3 grid((unsigned)(per_xcd * FI_XCDS), (unsigned)groups, 1);
    hipLaunchKernelGGL(fi_forward_ori_lds, grid, dim3(FI_THREADS, 1, 1), 0, (hipStream_t)stream, input1, input2,
                       input3, output, channel, h, w, s1, s2, s3, tiles_x, tiles_y, ntiles, per_xcd, ch_per_group);
    return launch_status();
}
