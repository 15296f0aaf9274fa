// correlation.hip -- PWC-Net / FlowNet cost volume for gfx950.
//
// Semantics: correlation_cuda_kernel.cu:47-147 (forward) and :151-334 (backward)
// of the reference, output-size math of correlation_cuda.cc:23-36; entry points
// replace correlation_cuda.cc.
//
//   out[b, (tj+dr)*dsz + (ti+dr), y, x] =
//       (1 / (k*k*C)) * sum_{j,i in kxk} sum_c P1[b,c,y1+j,x1+i] * P2[b,c,y1+tj*s2+j,x1+ti*s2+i]
//   with P = input zero-padded by pad_size, y1 = y*s1 + md, x1 likewise.
//
// The reference first repacks both inputs to zero-padded NHWC and then runs one
// 32-thread block per output pixel.  Here nothing is repacked: the padding is
// an index test, and for the configuration PWC-Net uses (k=1, s1=s2=1) a
// workgroup owns a 32x8 tile of output pixels, stages the matching window of
// the second feature map (tile + 2*md halo) in LDS one channel chunk at a
// time, and each lane keeps all (2*md+1)^2 running sums in registers.  The coarser pyramid
// levels (hundreds to a few thousand pixels, up to 196 channels) would leave most of the chip
// idle that way; they use 16x4-pixel tiles with one WAVE PER DISPLACEMENT ROW (9 waves per
// workgroup, 9 running sums per lane), which spreads the same work over 9x more waves.
#include "vfi_common.h"

#include <algorithm>
#include <type_traits>

#include <hip/hip_fp16.h>

namespace vfi {

// The tensors of a launch: one call, or two calls of equal shape in one launch (both flow directions of a pyramid level: at
// the coarse levels a launch is latency, 13-18 us for 1 MB, and two cost what one does).  Images 0 .. per - 1 are item 0's.
struct CorrItems { const float* in1[2]; const float* in2[2]; float* out[2]; int per; };

#define CORR_CC 8       // channels staged per LDS fill
#define CORR_CC_ROWS 8   // ... in the small-frame kernel (16 measured the same: its chunk loop is LDS-bound, one workgroup per CU)

__device__ __forceinline__ float padded_at(const float* __restrict__ f, int h, int w, int y, int x) {
    return (y >= 0 && y < h && x >= 0 && x < w) ? f[(int64_t)y * w + x] : 0.0f;
}

// k == 1, stride1 == stride2 == 1.  `org` = md - pad: output pixel (oy, ox) is
// centred on input pixel (oy + org, ox + org).
template <int MD, int CORR_TW, int CORR_TH>
__global__ __launch_bounds__(CORR_TW * CORR_TH) void corr_forward_k1(
    CorrItems items, int channel, int h, int w, int oh, int ow, int org) {
    constexpr int D = 2 * MD + 1;
    constexpr int LW = CORR_TW + 2 * MD, LH = CORR_TH + 2 * MD;
    constexpr int NI = (LH + CORR_TH - 1) / CORR_TH, NJ = (LW + CORR_TW - 1) / CORR_TW;
    __shared__ float tile[CORR_CC][LH][LW];

    const int tx = threadIdx.x, ty = threadIdx.y;
    const int ox = blockIdx.x * CORR_TW + tx, oy = blockIdx.y * CORR_TH + ty;
    const int item_ = (int)blockIdx.z >= items.per ? 1 : 0;   // (two calls in one launch: vfi_correlation_forward_pair)
    const int b = (int)blockIdx.z - item_ * items.per;
    const float* __restrict__ in1 = items.in1[item_];
    const float* __restrict__ in2 = items.in2[item_];
    float* __restrict__ out = items.out[item_];
    const int64_t plane = (int64_t)h * w;
    const float* f1 = in1 + (int64_t)b * channel * plane;
    const float* f2 = in2 + (int64_t)b * channel * plane;
    const int y1 = oy + org, x1 = ox + org;                 // centre in input coordinates
    const int wy0 = blockIdx.y * CORR_TH + org - MD;        // window origin in input coordinates
    const int wx0 = blockIdx.x * CORR_TW + org - MD;

    float acc[D * D];
#pragma unroll
    for (int k = 0; k < D * D; ++k) acc[k] = 0.0f;

    // staging plan of this thread, the same for every channel: window rows ty + i*TH, columns
    // tx + j*TW -- row segments, no index arithmetic per element; zero padding is the bounds test
    int soff[NI][NJ];
    bool sin[NI][NJ], sok[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = ty + i * CORR_TH, col = tx + j * CORR_TW;
            const int gy = wy0 + r, gx = wx0 + col;
            sin[i][j] = r < LH && col < LW;
            sok[i][j] = sin[i][j] && gy >= 0 && gy < h && gx >= 0 && gx < w;
            soff[i][j] = sok[i][j] ? gy * w + gx : 0;
        }
    const bool f1ok = y1 >= 0 && y1 < h && x1 >= 0 && x1 < w;
    const int f1off = f1ok ? y1 * w + x1 : 0;

    // the next chunk's values are fetched into registers before the current chunk is multiplied and
    // written to LDS after it: the finest level has ~2 workgroups per CU, too few to hide the
    // load -> LDS -> multiply chain otherwise
    float nv[CORR_CC][NI][NJ], na[CORR_CC];
    auto fetch = [&](int c0) {
        const int cn = min(CORR_CC, channel - c0);
#pragma unroll
        for (int c = 0; c < CORR_CC; ++c) {
            const float* p = f2 + (int64_t)(c0 + c) * plane;
            const bool cok = c < cn;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) nv[c][i][j] = (cok && sok[i][j]) ? p[soff[i][j]] : 0.0f;
            na[c] = (cok && f1ok) ? f1[(int64_t)(c0 + c) * plane + f1off] : 0.0f;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < channel; c0 += CORR_CC) {
        const int cn = min(CORR_CC, channel - c0);
        __syncthreads();
        float a[CORR_CC];
#pragma unroll
        for (int c = 0; c < CORR_CC; ++c) {
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if (sin[i][j]) tile[c][ty + i * CORR_TH][tx + j * CORR_TW] = nv[c][i][j];
            a[c] = na[c];
        }
        __syncthreads();
        if (c0 + CORR_CC < channel) fetch(c0 + CORR_CC);
        for (int c = 0; c < cn; ++c) {
            const float av = a[c];
#pragma unroll
            for (int tj = 0; tj < D; ++tj)
#pragma unroll
                for (int ti = 0; ti < D; ++ti)
                    acc[tj * D + ti] = fmaf(av, tile[c][ty + tj][tx + ti], acc[tj * D + ti]);
        }
    }
    if (ox < ow && oy < oh) {
        const float nelems = (float)channel;
        float* o = out + (int64_t)b * (D * D) * oh * ow + (int64_t)oy * ow + ox;
#pragma unroll
        for (int k = 0; k < D * D; ++k) o[(int64_t)k * oh * ow] = acc[k] / nelems;
    }
}

// k == 1, strides 1, small frames: tile of 16x4 output pixels, threadIdx.y = displacement row tj.
template <int MD>
__global__ __launch_bounds__(64 * (2 * MD + 1)) void corr_forward_k1_rows(
    CorrItems items, int channel, int h, int w, int oh, int ow, int org) {
    constexpr int D = 2 * MD + 1;
    constexpr int TW = 16, TH = 4, LW = TW + 2 * MD, LH = TH + 2 * MD;
    constexpr int NT = 64 * D, NE = CORR_CC_ROWS * LH * LW;      // threads, staged elements per chunk
    constexpr int NPT = (NE + NT - 1) / NT;
    __shared__ float tile[CORR_CC_ROWS][LH][LW];
    __shared__ float f1s[CORR_CC_ROWS][TH * TW];

    const int lane = threadIdx.x, tj = threadIdx.y;          // lane = pixel inside the tile
    const int tid = tj * 64 + lane;
    const int px = lane & (TW - 1), py = lane >> 4;
    const int ox = blockIdx.x * TW + px, oy = blockIdx.y * TH + py;
    const int item_ = (int)blockIdx.z >= items.per ? 1 : 0;   // (two calls in one launch: vfi_correlation_forward_pair)
    const int b = (int)blockIdx.z - item_ * items.per;
    const float* __restrict__ in1 = items.in1[item_];
    const float* __restrict__ in2 = items.in2[item_];
    float* __restrict__ out = items.out[item_];
    const int64_t plane = (int64_t)h * w;
    const float* f1 = in1 + (int64_t)b * channel * plane;
    const float* f2 = in2 + (int64_t)b * channel * plane;
    const int y1 = oy + org, x1 = ox + org;
    const int wy0 = blockIdx.y * TH + org - MD, wx0 = blockIdx.x * TW + org - MD;

    // staging plan: flat element e = tid + k*NT of the chunk's [CC][LH][LW] block (constant divisors)
    int soff[NPT], sch[NPT];
    bool sok[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int e = tid + k * NT;
        const int c = e / (LH * LW), rem = e - c * (LH * LW);
        const int r = rem / LW, col = rem - r * LW;
        const int gy = wy0 + r, gx = wx0 + col;
        sch[k] = c;
        sok[k] = e < NE && gy >= 0 && gy < h && gx >= 0 && gx < w;
        soff[k] = sok[k] ? gy * w + gx : 0;
    }
    const bool f1ok = y1 >= 0 && y1 < h && x1 >= 0 && x1 < w;
    const int f1off = f1ok ? y1 * w + x1 : 0;

    float acc[D];
#pragma unroll
    for (int ti = 0; ti < D; ++ti) acc[ti] = 0.0f;

    // The chunk loop is a dependent chain (global loads -> LDS -> products) and a small level has too
    // few workgroups to hide it: the next chunk's values are fetched into registers before the
    // current chunk is multiplied, and only written to LDS after it.
    constexpr int NF1 = (CORR_CC_ROWS + D - 1) / D;          // channels of the first map each wave stages
    float nv[NPT], nf1[NF1];
    auto fetch = [&](int c0) {
        const int cn = min(CORR_CC_ROWS, channel - c0);
#pragma unroll
        for (int k = 0; k < NPT; ++k)
            nv[k] = (sok[k] && sch[k] < cn) ? f2[(int64_t)(c0 + sch[k]) * plane + soff[k]] : 0.0f;
#pragma unroll
        for (int q = 0; q < NF1; ++q) {
            const int cc = tj + q * D;
            nf1[q] = (cc < cn && f1ok) ? f1[(int64_t)(c0 + cc) * plane + f1off] : 0.0f;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < channel; c0 += CORR_CC_ROWS) {
        const int cn = min(CORR_CC_ROWS, channel - c0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int e = tid + k * NT;
            if (e < NE) (&tile[0][0][0])[e] = nv[k];
        }
#pragma unroll
        for (int q = 0; q < NF1; ++q)
            if (tj + q * D < CORR_CC_ROWS) f1s[tj + q * D][lane] = nf1[q];
        __syncthreads();
        if (c0 + CORR_CC_ROWS < channel) fetch(c0 + CORR_CC_ROWS);
        for (int c = 0; c < cn; ++c) {
            const float av = f1s[c][lane];
#pragma unroll
            for (int ti = 0; ti < D; ++ti) acc[ti] = fmaf(av, tile[c][py + tj][px + ti], acc[ti]);
        }
    }
    if (ox < ow && oy < oh) {
        const float nelems = (float)channel;
        float* o = out + ((int64_t)b * (D * D) + tj * D) * oh * ow + (int64_t)oy * ow + ox;
#pragma unroll
        for (int ti = 0; ti < D; ++ti) o[(int64_t)ti * oh * ow] = acc[ti] / nelems;
    }
}

// k == 1, strides 1, frames whose rows are 16-byte aligned: tile of 32x4 output pixels,
// threadIdx.y = displacement row tj, a lane owns TWO horizontally adjacent pixels.  The window and the
// first map are staged as 16-byte units (global_load_dwordx4 -> ds_write_b128: a quarter of the
// vector-memory and LDS-write instructions of the one-float staging, which is what bounds the other
// tiled kernels at one workgroup per CU); per channel a lane reads the 10 values its two 9-wide
// displacement rows share as five aligned ds_read_b64 for 18 multiply-adds.  Same sequential
// channel order.
template <int MD>
__global__ __launch_bounds__(64 * (2 * MD + 1)) void corr_forward_k1_rows2(
    CorrItems items, int channel, int h, int w, int oh, int ow, int org) {
    constexpr int D = 2 * MD + 1;
    constexpr int TW = 32, TH = 4, LW = TW + 2 * MD, LH = TH + 2 * MD;          // LW = 40: 10 aligned 16-byte units
    constexpr int NT = 64 * D;
    constexpr int UW = LW / 4, NU = CORR_CC_ROWS * LH * UW;                      // staged 16-byte units per chunk
    constexpr int NPT = (NU + NT - 1) / NT;
    constexpr int FU = CORR_CC_ROWS * TH * (TW / 4);                             // ... of the first map
    constexpr int NF1 = (FU + NT - 1) / NT;
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float tile[CORR_CC_ROWS][LH][LW];
    __shared__ __attribute__((aligned(16))) float f1s[CORR_CC_ROWS][TH * TW];

    const int lane = threadIdx.x, tj = threadIdx.y;
    const int tid = tj * 64 + lane;
    const int px = 2 * (lane & 15), py = lane >> 4;
    const int ox = blockIdx.x * TW + px, oy = blockIdx.y * TH + py;
    const int item_ = (int)blockIdx.z >= items.per ? 1 : 0;   // (two calls in one launch: vfi_correlation_forward_pair)
    const int b = (int)blockIdx.z - item_ * items.per;
    const float* __restrict__ in1 = items.in1[item_];
    const float* __restrict__ in2 = items.in2[item_];
    float* __restrict__ out = items.out[item_];
    const int64_t plane = (int64_t)h * w;
    const float* f1 = in1 + (int64_t)b * channel * plane;
    const float* f2 = in2 + (int64_t)b * channel * plane;
    // window origin in input coordinates; a multiple of 4 columns (the host checks org and MD), so with
    // w a multiple of 4 every 16-byte unit lies wholly inside or wholly outside the frame
    const int wy0 = blockIdx.y * TH + org - MD, wx0 = blockIdx.x * TW + org - MD;

    // staging plans: unit e = tid + k*NT of the chunk's [CC][LH][UW] window block and of its [CC][TH][TW/4] first-map block
    // (constant divisors), as byte offsets from the chunk's first plane; the loads are buffer loads through a descriptor that
    // spans exactly the chunk's planes (corr_forward_k1_quad: units outside the frame and channels past the last return zeros)
    unsigned soff[NPT], foff[NF1];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int e = tid + k * NT;
        const int c = e / (LH * UW), rem = e - c * (LH * UW);
        const int r = rem / UW, col = 4 * (rem - r * UW);
        const int gy = wy0 + r, gx = wx0 + col;
        const bool ok = e < NU && gy >= 0 && gy < h && gx >= 0 && gx < w;
        soff[k] = ok ? 4u * (unsigned)(c * (int)plane + gy * w + gx) : 0x80000000u;
    }
#pragma unroll
    for (int k = 0; k < NF1; ++k) {
        const int e = tid + k * NT;
        const int c = e / (TH * (TW / 4)), rem = e - c * (TH * (TW / 4));
        const int gy = blockIdx.y * TH + rem / (TW / 4) + org, gx = blockIdx.x * TW + 4 * (rem % (TW / 4)) + org;
        const bool ok = e < FU && gy >= 0 && gy < h && gx >= 0 && gx < w;
        foff[k] = ok ? 4u * (unsigned)(c * (int)plane + gy * w + gx) : 0x80000000u;
    }

    float acc[2][D];
#pragma unroll
    for (int ti = 0; ti < D; ++ti) { acc[0][ti] = 0.0f; acc[1][ti] = 0.0f; }

    // the next chunk's units are fetched into registers before the current chunk is multiplied
    v4f nv[NPT], nf[NF1];
    auto fetch = [&](int c0) {
        const int cn = min(CORR_CC_ROWS, channel - c0);
        const int bytes = cn * (int)plane * 4;
        const auto d2 = __builtin_amdgcn_make_buffer_rsrc((void*)(f2 + (int64_t)c0 * plane), 0, bytes, 0x00020000);
        const auto d1 = __builtin_amdgcn_make_buffer_rsrc((void*)(f1 + (int64_t)c0 * plane), 0, bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NPT; ++k) nv[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(d2, soff[k], 0, 0));
#pragma unroll
        for (int k = 0; k < NF1; ++k) nf[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(d1, foff[k], 0, 0));
    };
    fetch(0);
    for (int c0 = 0; c0 < channel; c0 += CORR_CC_ROWS) {
        const int cn = min(CORR_CC_ROWS, channel - c0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int e = tid + k * NT;
            if (e < NU) reinterpret_cast<v4f*>(&tile[0][0][0])[e] = nv[k];
        }
#pragma unroll
        for (int k = 0; k < NF1; ++k) {
            const int e = tid + k * NT;
            if (e < FU) reinterpret_cast<v4f*>(&f1s[0][0])[e] = nf[k];
        }
        __syncthreads();
        if (c0 + CORR_CC_ROWS < channel) fetch(c0 + CORR_CC_ROWS);
        for (int c = 0; c < cn; ++c) {
            const v2f a = *reinterpret_cast<const v2f*>(&f1s[c][py * TW + px]);
            const v2f* row = reinterpret_cast<const v2f*>(&tile[c][py + tj][px]);
            float t[D + 1];
#pragma unroll
            for (int k = 0; k < (D + 1) / 2; ++k) {
                const v2f q = row[k];
                t[2 * k] = q.x;
                t[2 * k + 1] = q.y;
            }
            // (one v_fmac per term: left to itself the compiler packs the two pixels' terms into v_pk_fma_f32 -- 1.6 x a plain
            //  multiply-add each on gfx950 -- and pays four v_pk_mov per channel to line up the odd operand pairs: no faster)
#pragma unroll
            for (int ti = 0; ti < D; ++ti) {
                asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[0][ti]) : "v"(a.x), "v"(t[ti]));
                asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[1][ti]) : "v"(a.y), "v"(t[ti + 1]));
            }
        }
    }
    // (the mean of a power-of-two channel count as a product with the exact reciprocal: corr_forward_k1_quad)
    const float nelems = (float)channel;
    const bool pow2 = (channel & (channel - 1)) == 0;
    const float inv = 1.0f / nelems;
    auto store = [&](auto POW2) {
        auto mean = [&](float v) { return decltype(POW2)::value ? v * inv : v / nelems; };
        if (oy < oh && ox + 1 < ow && (ow & 1) == 0) {
            // the lane's two pixels as one 8-byte store: a wave writes whole 128-byte row segments
            float* o = out + ((int64_t)b * (D * D) + tj * D) * oh * ow + (int64_t)oy * ow + ox;
#pragma unroll
            for (int ti = 0; ti < D; ++ti) *reinterpret_cast<v2f*>(o + (int64_t)ti * oh * ow) = v2f{mean(acc[0][ti]), mean(acc[1][ti])};
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (ox + q < ow && oy < oh) {
                    float* o = out + ((int64_t)b * (D * D) + tj * D) * oh * ow + (int64_t)oy * ow + ox + q;
#pragma unroll
                    for (int ti = 0; ti < D; ++ti) o[(int64_t)ti * oh * ow] = mean(acc[q][ti]);
                }
        }
    };
    if (pow2) store(std::true_type{}); else store(std::false_type{});
}

// k == 1, strides 1, 16-byte-aligned rows, the LARGE levels (the finest pyramid level at 1080p: 288 x 496).  With one wave per
// displacement row a workgroup is 9 waves, three fit a CU, and the finest level's 1,152 tiles of corr_forward_k1_rows2 run as
// one full round and one half-empty one (33.7 us; the pair of both directions, exactly three rounds: 62.7).  Here a workgroup
// is 3 waves and owns THREE displacement rows (tj = 3 g + wave) of a 64 x 4 tile, a lane owns FOUR horizontally adjacent
// pixels: 36 running sums per lane, and per channel one 16-byte read of the first map and three of the second map's row feed
// 36 multiply-adds (rows2: six 8-byte reads for 18).  The window a workgroup stages is the tile + 8 columns x 6 rows (the three
// displacement rows its waves need) -- the three workgroups of a tile run on one XCD back to back and find each other's
// lines in its L2.  ~7 workgroups per CU are resident: every tile of the finest level at once.  Channels in sequence, the same
// fmaf per term: the same bits as every other kernel here.
//
// Staging: four channels per chunk by LDS-DMA (`buffer_load_dwordx4 ... lds`) into TWO LDS buffers.  The window and first-map
// units are buffer loads through a descriptor that spans exactly the chunk's planes, with one constant byte offset per unit
// (a unit outside the frame has an offset out of any range, a channel past the last one falls out of the descriptor's: both
// arrive as zeros); a chunk's units go from memory straight into the buffer the previous chunk was read from, under the current
// chunk's multiply-adds -- no staging registers, no ds_write, no address arithmetic, one barrier per chunk.  (The round's first
// version fetched the units into registers and wrote them to LDS: 87 registers, two barriers per chunk, 24 us; this one 63
// registers, 20.7 us at 32 x 288 x 496.)  The tap reads are asm: hipcc
// drains vmcnt before an LDS read it can see next to an LDS-DMA target (filterinterp_lds.hip), which would wait for the chunk
// in flight.  Same tile, same lanes, same order of the multiply-adds: the same bits.
typedef __attribute__((address_space(3))) void* corr_lptr_t;
#define CORR_QUAD_LDS_FLOATS (2 * (4 * 6 * 18 + 4 * 4 * 16) * 4)          // two buffers of 432 window units + 256 first-map units
template <int MD>
__device__ __forceinline__ void corr_quad_body(
    const CorrItems& items, int channel, int h, int w, int oh, int ow, int org, int tiles_x, int tiles_y, int ntiles, float* lds) {
    constexpr int D = 2 * MD + 1, G = 3;
    constexpr int CCQ = 4;
    constexpr int TW = 64, TH = 4, LW = TW + 2 * MD, LH = TH + G - 1;
    constexpr int NT = 64 * G;
    constexpr int UW = LW / 4, NU = CCQ * LH * UW;                                // 432 window units (16 bytes) per chunk
    constexpr int NPT = (NU + NT - 1) / NT;
    constexpr int FU = CCQ * TH * (TW / 4);                                       // 256 first-map units
    constexpr int NF1 = (FU + NT - 1) / NT;
    constexpr int BUF = (NU + FU) * 4;                                            // floats per buffer: [window units][first-map units]
    typedef float v4f __attribute__((ext_vector_type(4)));
    static_assert(2 * BUF == CORR_QUAD_LDS_FLOATS, "the kernel's LDS array");

    const int xcd = blockIdx.x % 8, kq = blockIdx.x / 8;
    const int g = kq % (D / G), t_ = (kq / (D / G)) * 8 + xcd;
    if (t_ >= ntiles) return;
    const int img_ = t_ / (tiles_x * tiles_y), trem = t_ - img_ * (tiles_x * tiles_y);
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int lane = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int tid = wv * 64 + lane;
    const int tj = G * g + wv;
    const int px = 4 * (lane & 15), py = lane >> 4;
    const int ox = txi * TW + px, oy = tyi * TH + py;
    const int item_ = img_ >= items.per ? 1 : 0;
    const int b = img_ - item_ * items.per;
    const float* __restrict__ in1 = items.in1[item_];
    const float* __restrict__ in2 = items.in2[item_];
    float* __restrict__ out = items.out[item_];
    const int64_t plane = (int64_t)h * w;
    const float* f1 = in1 + (int64_t)b * channel * plane;
    const float* f2 = in2 + (int64_t)b * channel * plane;
    const int wy0 = tyi * TH + org - MD + G * g, wx0 = txi * TW + org - MD;

    // byte offsets of this thread's units from the chunk's first plane (out of any range: a unit outside the frame)
    unsigned soff[NPT], foff[NF1];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int e = tid + k * NT;
        const int c = e / (LH * UW), rem = e - c * (LH * UW);
        const int r = rem / UW, col = 4 * (rem - r * UW);
        const int gy = wy0 + r, gx = wx0 + col;
        const bool ok = e < NU && gy >= 0 && gy < h && gx >= 0 && gx < w;
        soff[k] = ok ? 4u * (unsigned)(c * (int)plane + gy * w + gx) : 0x80000000u;
    }
#pragma unroll
    for (int k = 0; k < NF1; ++k) {
        const int e = tid + k * NT;
        const int c = e / (TH * (TW / 4)), rem = e - c * (TH * (TW / 4));
        const int gy = tyi * TH + rem / (TW / 4) + org, gx = txi * TW + 4 * (rem % (TW / 4)) + org;
        const bool ok = e < FU && gy >= 0 && gy < h && gx >= 0 && gx < w;
        foff[k] = ok ? 4u * (unsigned)(c * (int)plane + gy * w + gx) : 0x80000000u;
    }
    // a wave's 64 lanes write 64 consecutive units: wave-uniform destination + lane * 16 bytes (M0).  A staging instruction whose
    // units lie past the block's end is skipped by the waves it has nothing for and runs with a partial exec mask in the last one.
    auto issue = [&](int c0, int buf) {
        const int cn = min(CCQ, channel - c0);
        const int bytes = cn * (int)plane * 4;
        const auto d2 = __builtin_amdgcn_make_buffer_rsrc((void*)(f2 + (int64_t)c0 * plane), 0, bytes, 0x00020000);
        const auto d1 = __builtin_amdgcn_make_buffer_rsrc((void*)(f1 + (int64_t)c0 * plane), 0, bytes, 0x00020000);
        float* base = lds + buf * BUF;
#pragma unroll
        for (int k = 0; k < NPT; ++k)
            if (tid + k * NT < NU)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(d2, (corr_lptr_t)(base + (k * NT + wv * 64) * 4), 16, soff[k], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NF1; ++k)
            if (tid + k * NT < FU)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(d1, (corr_lptr_t)(base + NU * 4 + (k * NT + wv * 64) * 4), 16, foff[k], 0, 0, 0);
    };

    float acc[4][D];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ti = 0; ti < D; ++ti) acc[q][ti] = 0.0f;

    const unsigned lds0 = (unsigned)(uintptr_t)(corr_lptr_t)lds;
    const unsigned a_addr = lds0 + 4u * (unsigned)(NU * 4 + py * TW + px);                 // first map: [c][TH * TW]
    const unsigned t_addr = lds0 + 4u * (unsigned)((py + wv) * LW + px);                   // window: [c][LH][LW]
#define CORR_CHANNEL_TERMS(c, bo) do { \
        v4f a4, r0, r1, r2; \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a4) : "v"(a_addr + (bo)), "n"((c) * TH * TW * 4)); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r0) : "v"(t_addr + (bo)), "n"((c) * LH * LW * 4)); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r1) : "v"(t_addr + (bo)), "n"((c) * LH * LW * 4 + 16)); \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r2) : "v"(t_addr + (bo)), "n"((c) * LH * LW * 4 + 32)); \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4), "+v"(r0), "+v"(r1), "+v"(r2)); \
        const float a_[4] = { a4.x, a4.y, a4.z, a4.w }; \
        const float t_v[12] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w }; \
        _Pragma("unroll") for (int ti = 0; ti < D; ++ti) \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc[q][ti]) : "v"(a_[q]), "v"(t_v[q + ti])); \
    } while (0)

    issue(0, 0);
    int buf = 0;
    for (int c0 = 0; c0 < channel; c0 += CCQ) {
        const int cn = min(CCQ, channel - c0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's units of the chunk have landed ...
        __builtin_amdgcn_s_barrier();                           // ... everybody's have, and everybody is done with the other buffer
        if (c0 + CCQ < channel) issue(c0 + CCQ, buf ^ 1);
        const unsigned bo = (unsigned)(buf * BUF * 4);
        CORR_CHANNEL_TERMS(0, bo);
        if (cn > 1) CORR_CHANNEL_TERMS(1, bo);
        if (cn > 2) CORR_CHANNEL_TERMS(2, bo);
        if (cn > 3) CORR_CHANNEL_TERMS(3, bo);
        buf ^= 1;
    }
#undef CORR_CHANNEL_TERMS
    static_assert(CCQ == 4, "four channels per chunk");
    if (oy >= oh) return;
    float* o = out + ((int64_t)b * (D * D) + tj * D) * oh * ow + (int64_t)oy * ow + ox;
    const float nelems = (float)channel;
    const bool pow2 = (channel & (channel - 1)) == 0;
    const float inv = 1.0f / nelems;
    auto store = [&](auto POW2) {
        auto mean = [&](float v) { return decltype(POW2)::value ? v * inv : v / nelems; };
        if (ox + 3 < ow && (ow & 3) == 0) {
#pragma unroll
            for (int ti = 0; ti < D; ++ti)
                *reinterpret_cast<v4f*>(o + (int64_t)ti * oh * ow) = v4f{mean(acc[0][ti]), mean(acc[1][ti]), mean(acc[2][ti]), mean(acc[3][ti])};
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ox + q < ow) {
#pragma unroll
                    for (int ti = 0; ti < D; ++ti) o[(int64_t)ti * oh * ow + q] = mean(acc[q][ti]);
                }
        }
    };
    if (pow2) store(std::true_type{}); else store(std::false_type{});
}

// (the body is a device function: its inline assembly must not be instantiated by the host pass)
template <int MD>
__global__ __launch_bounds__(192, 5) void corr_forward_k1_quad(
    CorrItems items, int channel, int h, int w, int oh, int ow, int org, int tiles_x, int tiles_y, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[CORR_QUAD_LDS_FLOATS];  // ONE array (see above)
    corr_quad_body<MD>(items, channel, h, w, oh, ow, org, tiles_x, tiles_y, ntiles, lds);
}

// (explicit: hipcc 7.2 leaves the host stub of a kernel template undefined when its only launch sits in a branch it folds away)
template __global__ void corr_forward_k1_quad<4>(CorrItems, int, int, int, int, int, int, int, int, int);

#ifdef VFI_DEV
// DEVELOPMENT BUILDS ONLY (measured, slower: see the end of this comment).
// k == 1, strides 1, md == 4 on the matrix cores (round 3; BASELINE.json north_star: "the correlation cost-volume is the
// one dense-contraction candidate for MFMA").  For one output row y and one displacement row tj the cost volume is the
// band x2 - x1 in [-4, 4] of the product  C[x1][x2] = sum_c f1[c][y][x1] * f2[c][y + tj - 4][x2]:  a wave owns 16
// pixels x1 of a row and forms, per displacement row, the two 16x16 tiles that hold its band (x2 in [x0 - 4, x0 + 28))
// with v_mfma_f32_16x16x4_f32, four channels per instruction -- 18 accumulator tiles (72 registers) per wave, 144 of
// every 512 products wanted.  The f32 MFMA is bit for bit a k-ordered fmaf chain (cdna_hip_programming.md section 3),
// so the sums are the sequential-channel-order sums of the other kernels: same bits.  A workgroup of 16 waves owns
// 64x4 pixels; both maps are staged eight channels at a time as 16-byte units exactly as in corr_forward_k1_rows2.
// Per four channels a wave reads 1 + 18 operand registers from LDS (ds_read_b32) for 18 MFMAs.
// Measured on the 1080p pyramid (tools/corr_mfma_ab.py): bit-identical with the vector kernels on every level, and slower --
// 32 x 288x496: 57.0 us against 38.7 us, 64 x 144x248: 31.1 against 24.0, 96 x 72x124: 39.8 against 19.4 (62 workgroups of 16
// waves do not fill the chip).  The f32 MFMA runs at the vector units' rate and 72 % of its products fall outside the band.
template <int MD>
__global__ __launch_bounds__(1024) void corr_forward_k1_mfma(
    CorrItems items, int channel, int h, int w, int oh, int ow, int org) {
    static_assert(MD == 4, "two 16-wide tiles hold a band of 9 around 16 pixels");
    constexpr int D = 2 * MD + 1;
    constexpr int TW = 64, TH = 4, LW = TW + 16, LH = TH + 2 * MD;              // window: 80 columns (20 units), 12 rows
    constexpr int CC = 8, NT = 1024;
    constexpr int UW = LW / 4, NU = CC * LH * UW, NPT = (NU + NT - 1) / NT;      // 1920 units: 2 per thread
    constexpr int FU = CC * TH * (TW / 4);                                       // 512 units of the first map
    typedef float v4f __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float tile[CC][LH][LW];
    __shared__ __attribute__((aligned(16))) float f1s[CC][TH][TW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int seg = wave & 3, ry = wave >> 2;                                    // 16-pixel segment and row of the tile
    const int item_ = (int)blockIdx.z >= items.per ? 1 : 0;   // (two calls in one launch: vfi_correlation_forward_pair)
    const int b = (int)blockIdx.z - item_ * items.per;
    const float* __restrict__ in1 = items.in1[item_];
    const float* __restrict__ in2 = items.in2[item_];
    float* __restrict__ out = items.out[item_];
    const int64_t plane = (int64_t)h * w;
    const float* f1 = in1 + (int64_t)b * channel * plane;
    const float* f2 = in2 + (int64_t)b * channel * plane;
    const int wy0 = blockIdx.y * TH + org - MD, wx0 = blockIdx.x * TW + org - MD;   // window origin (a multiple of 4 columns)

    int soff[NPT], sch[NPT];
    bool sok[NPT];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int e = tid + k * NT;
        const int c = e / (LH * UW), rem = e - c * (LH * UW);
        const int r = rem / UW, col = 4 * (rem - r * UW);
        const int gy = wy0 + r, gx = wx0 + col;
        sch[k] = c;
        sok[k] = e < NU && gy >= 0 && gy < h && gx >= 0 && gx < w;
        soff[k] = sok[k] ? gy * w + gx : 0;
    }
    int foff, fch;
    bool fok;
    {
        const int e = tid;
        const int c = e / (TH * (TW / 4)), rem = e - c * (TH * (TW / 4));
        const int gy = blockIdx.y * TH + rem / (TW / 4) + org, gx = blockIdx.x * TW + 4 * (rem % (TW / 4)) + org;
        fch = c;
        fok = e < FU && gy >= 0 && gy < h && gx >= 0 && gx < w;
        foff = fok ? gy * w + gx : 0;
    }

    v4f acc[D][2];
#pragma unroll
    for (int tj = 0; tj < D; ++tj) { acc[tj][0] = v4f{0.0f, 0.0f, 0.0f, 0.0f}; acc[tj][1] = v4f{0.0f, 0.0f, 0.0f, 0.0f}; }

    v4f nv[NPT], nf;
    const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
    auto fetch = [&](int c0) {
        const int cn = min(CC, channel - c0);
#pragma unroll
        for (int k = 0; k < NPT; ++k)
            nv[k] = (sok[k] && sch[k] < cn) ? *reinterpret_cast<const v4f*>(f2 + (int64_t)(c0 + sch[k]) * plane + soff[k]) : zero;
        nf = (fok && fch < cn) ? *reinterpret_cast<const v4f*>(f1 + (int64_t)(c0 + fch) * plane + foff) : zero;
    };
    fetch(0);
    const int kk = lane >> 4, jj = lane & 15;                                    // operand element: channel kk of the step, column jj
    for (int c0 = 0; c0 < channel; c0 += CC) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int e = tid + k * NT;
            if (e < NU) reinterpret_cast<v4f*>(&tile[0][0][0])[e] = nv[k];
        }
        if (tid < FU) reinterpret_cast<v4f*>(&f1s[0][0][0])[tid] = nf;
        __syncthreads();
        if (c0 + CC < channel) fetch(c0 + CC);
        // (channels past the end were staged as zeros: they add exact zeros)
#pragma unroll
        for (int ks = 0; ks < CC / 4; ++ks) {
            const float a = f1s[4 * ks + kk][ry][16 * seg + jj];
#pragma unroll
            for (int tj = 0; tj < D; ++tj) {
                const float b0 = tile[4 * ks + kk][ry + tj][16 * seg + jj];
                const float b1 = tile[4 * ks + kk][ry + tj][16 * seg + 16 + jj];
                acc[tj][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[tj][0], 0, 0, 0);
                acc[tj][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[tj][1], 0, 0, 0);
            }
        }
    }
    // the band: out[tj][ti][x1] = C_tj[x1][x1 + ti] (column 0 of tile 0 is x2 = x0 - 4).  Lane l, register r of a tile hold
    // row 4 (l >> 4) + r, column l & 15: through an LDS image of the wave's two tiles, [16 rows][32 + 1 columns]
    __syncthreads();
    float* img = &tile[0][0][0] + wave * (16 * 33);                              // 16 waves x 528 floats < the window array
    const float nelems = (float)channel;
    const int oy = blockIdx.y * TH + ry, ox0 = blockIdx.x * TW + 16 * seg;
#pragma unroll
    for (int tj = 0; tj < D; ++tj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            img[(4 * (lane >> 4) + r) * 33 + (lane & 15)] = acc[tj][0][r];
            img[(4 * (lane >> 4) + r) * 33 + 16 + (lane & 15)] = acc[tj][1][r];
        }
        // (a wave's own LDS writes are visible to its reads in order: no barrier)
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int ti = 4 * pass + (lane >> 4), i = lane & 15;
            if (ti < D && oy < oh && ox0 + i < ow)
                out[((int64_t)b * (D * D) + tj * D + ti) * oh * ow + (int64_t)oy * ow + ox0 + i] = img[i * 33 + i + ti] / nelems;
        }
    }
}
#endif  // VFI_DEV

// k == 1, strides 1, tiny frames (the coarsest pyramid levels: a few hundred pixels, up to 196
// channels): one thread per output element, x fastest, so a wave reads 64 consecutive pixels of each
// map; eight channels in flight per thread.  A tiled kernel leaves most of the chip idle here (10
// workgroups for 18x31).  Sequential channel order, as everywhere.
template <int MD>
__global__ __launch_bounds__(256) void corr_forward_k1_flat(
    CorrItems items, int batch, int channel, int h, int w, int oh, int ow, int org) {
    constexpr int D = 2 * MD + 1;
    const int64_t total = (int64_t)batch * D * D * oh * ow;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int ox = (int)(gid % ow);
    const int oy = (int)((gid / ow) % oh);
    const int tc = (int)((gid / ((int64_t)ow * oh)) % (D * D));
    const int bz = (int)(gid / ((int64_t)ow * oh * D * D));
    const int item_ = bz >= items.per ? 1 : 0;              // (two calls in one launch: vfi_correlation_forward_pair)
    const int b = bz - item_ * items.per;
    const float* __restrict__ in1 = items.in1[item_];
    const float* __restrict__ in2 = items.in2[item_];
    float* __restrict__ out = items.out[item_];
    const int y1 = oy + org, x1 = ox + org;
    const int y2 = y1 + tc / D - MD, x2 = x1 + tc % D - MD;
    float acc = 0.0f;
    if (y1 >= 0 && y1 < h && x1 >= 0 && x1 < w && y2 >= 0 && y2 < h && x2 >= 0 && x2 < w) {     // else zero padding
        const int64_t plane = (int64_t)h * w;
        const float* p1 = in1 + (int64_t)b * channel * plane + (int64_t)y1 * w + x1;
        const float* p2 = in2 + (int64_t)b * channel * plane + (int64_t)y2 * w + x2;
        int c = 0;
        for (; c + 8 <= channel; c += 8) {
            float a[8], v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { a[k] = p1[(int64_t)(c + k) * plane]; v[k] = p2[(int64_t)(c + k) * plane]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = fmaf(a[k], v[k], acc);
        }
        for (; c < channel; ++c) acc = fmaf(p1[(int64_t)c * plane], p2[(int64_t)c * plane], acc);
    }
    out[gid - (int64_t)item_ * items.per * D * D * oh * ow] = acc / (float)channel;
}

// any kernel size / strides: one thread per output element, sequential channel order
__global__ __launch_bounds__(256) void corr_forward_generic(
    const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out,
    int batch, int channel, int h, int w, int oc, int oh, int ow,
    int pad, int kr, int md, int s1, int s2, int dr) {
    const int64_t total = (int64_t)batch * oc * oh * ow;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int ox = (int)(gid % ow);
    const int oy = (int)((gid / ow) % oh);
    const int tc = (int)((gid / ((int64_t)ow * oh)) % oc);
    const int b = (int)(gid / ((int64_t)ow * oh * oc));
    const int dsz = 2 * dr + 1;
    const int ti = tc % dsz - dr, tj = tc / dsz - dr;
    const int y1 = oy * s1 + md - pad, x1 = ox * s1 + md - pad;     // padded -> input coordinates
    const int y2 = y1 + tj * s2, x2 = x1 + ti * s2;
    const int64_t plane = (int64_t)h * w;
    const float* f1 = in1 + (int64_t)b * channel * plane;
    const float* f2 = in2 + (int64_t)b * channel * plane;
    float acc = 0.0f;
    for (int j = -kr; j <= kr; ++j)
        for (int i = -kr; i <= kr; ++i)
            for (int c = 0; c < channel; ++c)
                acc = fmaf(padded_at(f1 + (int64_t)c * plane, h, w, y1 + j, x1 + i),
                           padded_at(f2 + (int64_t)c * plane, h, w, y2 + j, x2 + i), acc);
    const int k = 2 * kr + 1;
    out[gid] = acc / (float)(k * k * channel);
}

// The at::Half instantiation for PWC-Net's configuration, tiled like corr_forward_k1_rows2: 32x4 output pixels per
// workgroup, one wave per displacement row, a lane owns two adjacent pixels.  Both maps are staged as halves
// (8-byte units); per channel and displacement ONE packed multiply forms the two pixels' products, each rounded to
// half exactly as the reference's `rInput1[i] * rInput2[i]` on two Half values (correlation_cuda_kernel.cu:124), and
// two mixed-precision adds accumulate them in float, channels in sequence.  Same bits as corr_forward_generic_f16.
template <int MD>
__global__ __launch_bounds__(64 * (2 * MD + 1)) void corr_forward_k1_rows2_f16(
    const __half* __restrict__ in1, const __half* __restrict__ in2, __half* __restrict__ out,
    int channel, int h, int w, int oh, int ow, int org) {
    constexpr int D = 2 * MD + 1;
    constexpr int TW = 32, TH = 4, LW = TW + 2 * MD, LH = TH + 2 * MD;          // LW = 40 halves: 10 aligned 8-byte units
    constexpr int NT = 64 * D;
    constexpr int UW = LW / 4, NU = CORR_CC_ROWS * LH * UW;
    constexpr int NPT = (NU + NT - 1) / NT;
    constexpr int FU = CORR_CC_ROWS * TH * (TW / 4);
    constexpr int NF1 = (FU + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) __half tile[CORR_CC_ROWS][LH][LW];
    __shared__ __attribute__((aligned(16))) __half f1s[CORR_CC_ROWS][TH * TW];

    const int lane = threadIdx.x, tj = threadIdx.y;
    const int tid = tj * 64 + lane;
    const int px = 2 * (lane & 15), py = lane >> 4;
    const int ox = blockIdx.x * TW + px, oy = blockIdx.y * TH + py;
    const int b = blockIdx.z;
    const int64_t plane = (int64_t)h * w;
    const __half* f1 = in1 + (int64_t)b * channel * plane;
    const __half* f2 = in2 + (int64_t)b * channel * plane;
    const int wy0 = blockIdx.y * TH + org - MD, wx0 = blockIdx.x * TW + org - MD;

    // staging plans as byte offsets from the chunk's first plane; buffer loads through a descriptor that spans exactly the chunk's
    // planes (corr_forward_k1_quad: units outside the frame and channels past the last one arrive as zeros)
    unsigned soff[NPT], foff[NF1];
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int e = tid + k * NT;
        const int c = e / (LH * UW), rem = e - c * (LH * UW);
        const int r = rem / UW, col = 4 * (rem - r * UW);
        const int gy = wy0 + r, gx = wx0 + col;
        const bool ok = e < NU && gy >= 0 && gy < h && gx >= 0 && gx < w;
        soff[k] = ok ? 2u * (unsigned)(c * (int)plane + gy * w + gx) : 0x80000000u;
    }
#pragma unroll
    for (int k = 0; k < NF1; ++k) {
        const int e = tid + k * NT;
        const int c = e / (TH * (TW / 4)), rem = e - c * (TH * (TW / 4));
        const int gy = blockIdx.y * TH + rem / (TW / 4) + org, gx = blockIdx.x * TW + 4 * (rem % (TW / 4)) + org;
        const bool ok = e < FU && gy >= 0 && gy < h && gx >= 0 && gx < w;
        foff[k] = ok ? 2u * (unsigned)(c * (int)plane + gy * w + gx) : 0x80000000u;
    }

    float acc[2][D];
#pragma unroll
    for (int ti = 0; ti < D; ++ti) { acc[0][ti] = 0.0f; acc[1][ti] = 0.0f; }

    uint2 nv[NPT], nf[NF1];
    typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
    auto fetch = [&](int c0) {
        const int cn = min(CORR_CC_ROWS, channel - c0);
        const int bytes = cn * (int)plane * 2;
        const auto d2 = __builtin_amdgcn_make_buffer_rsrc((void*)(f2 + (int64_t)c0 * plane), 0, bytes, 0x00020000);
        const auto d1 = __builtin_amdgcn_make_buffer_rsrc((void*)(f1 + (int64_t)c0 * plane), 0, bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NPT; ++k) { const v2u_ v = __builtin_amdgcn_raw_buffer_load_b64(d2, soff[k], 0, 0); nv[k] = make_uint2(v.x, v.y); }
#pragma unroll
        for (int k = 0; k < NF1; ++k) { const v2u_ v = __builtin_amdgcn_raw_buffer_load_b64(d1, foff[k], 0, 0); nf[k] = make_uint2(v.x, v.y); }
    };
    fetch(0);
    for (int c0 = 0; c0 < channel; c0 += CORR_CC_ROWS) {
        const int cn = min(CORR_CC_ROWS, channel - c0);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int e = tid + k * NT;
            if (e < NU) reinterpret_cast<uint2*>(&tile[0][0][0])[e] = nv[k];
        }
#pragma unroll
        for (int k = 0; k < NF1; ++k) {
            const int e = tid + k * NT;
            if (e < FU) reinterpret_cast<uint2*>(&f1s[0][0])[e] = nf[k];
        }
        __syncthreads();
        if (c0 + CORR_CC_ROWS < channel) fetch(c0 + CORR_CC_ROWS);
        for (int c = 0; c < cn; ++c) {
            const __half2 a = *reinterpret_cast<const __half2*>(&f1s[c][py * TW + px]);
            const __half2* row = reinterpret_cast<const __half2*>(&tile[c][py + tj][px]);
            __half2 r[(D + 1) / 2];
#pragma unroll
            for (int k = 0; k < (D + 1) / 2; ++k) r[k] = row[k];
#pragma unroll
            for (int ti = 0; ti < D; ++ti) {
                // (t[ti], t[ti + 1]): pixel 0 meets displacement column ti, pixel 1 the next one
                const __half2 pair = (ti & 1) ? __halves2half2(__high2half(r[ti / 2]), __low2half(r[ti / 2 + 1])) : r[ti / 2];
                const __half2 p = __hmul2(a, pair);                        // two products, each rounded to half
                // acc + (float)product in one mixed-precision instruction (fma(p, 1, acc): one rounding, the float add's)
                const unsigned pb = __builtin_bit_cast(unsigned, p);
                asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[0][ti]) : "v"(pb));
                asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[1][ti]) : "v"(pb));
            }
        }
    }
    // (the mean of a power-of-two channel count as a product with the exact reciprocal: the same real number, rounded to half once)
    const float nelems = (float)channel;
    const bool pow2 = (channel & (channel - 1)) == 0;
    const float inv = 1.0f / nelems;
    auto store = [&](auto POW2) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (ox + q < ow && oy < oh) {
                __half* o = out + ((int64_t)b * (D * D) + tj * D) * oh * ow + (int64_t)oy * ow + ox + q;
#pragma unroll
                for (int ti = 0; ti < D; ++ti) o[(int64_t)ti * oh * ow] = __float2half_rn(decltype(POW2)::value ? acc[q][ti] * inv : acc[q][ti] / nelems);
            }
    };
    if (pow2) store(std::true_type{}); else store(std::false_type{});
}

// half inputs and output, the reference's `scalar_t = at::Half` instantiation (correlation_cuda_kernel.cu:386,403):
// each product is formed in half (`rInput1[i] * rInput2[i]` on two Half values: one rounding to half, :124),
// widened and accumulated in float; the mean is rounded to half once (:143).  One thread per output element,
// sequential channel order, eight channels of loads in flight.
__global__ __launch_bounds__(256) void corr_forward_generic_f16(
    const __half* __restrict__ in1, const __half* __restrict__ in2, __half* __restrict__ out,
    int batch, int channel, int h, int w, int oc, int oh, int ow,
    int pad, int kr, int md, int s1, int s2, int dr) {
    const int64_t total = (int64_t)batch * oc * oh * ow;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int ox = (int)(gid % ow);
    const int oy = (int)((gid / ow) % oh);
    const int tc = (int)((gid / ((int64_t)ow * oh)) % oc);
    const int b = (int)(gid / ((int64_t)ow * oh * oc));
    const int dsz = 2 * dr + 1;
    const int ti = tc % dsz - dr, tj = tc / dsz - dr;
    const int y1 = oy * s1 + md - pad, x1 = ox * s1 + md - pad;     // padded -> input coordinates
    const int y2 = y1 + tj * s2, x2 = x1 + ti * s2;
    const int64_t plane = (int64_t)h * w;
    const __half* f1 = in1 + (int64_t)b * channel * plane;
    const __half* f2 = in2 + (int64_t)b * channel * plane;
    float acc = 0.0f;
    for (int j = -kr; j <= kr; ++j)
        for (int i = -kr; i <= kr; ++i) {
            const int ya = y1 + j, xa = x1 + i, yb = y2 + j, xb = x2 + i;
            // a tap outside either map multiplies a zero of the padding: +0 for every channel
            if (ya < 0 || ya >= h || xa < 0 || xa >= w || yb < 0 || yb >= h || xb < 0 || xb >= w) continue;
            const __half* pa = f1 + (int64_t)ya * w + xa;
            const __half* pb = f2 + (int64_t)yb * w + xb;
            int c = 0;
            for (; c + 8 <= channel; c += 8) {
                __half va[8], vb[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { va[q] = pa[(int64_t)(c + q) * plane]; vb[q] = pb[(int64_t)(c + q) * plane]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += __half2float(__hmul(va[q], vb[q]));
            }
            for (; c < channel; ++c) acc += __half2float(__hmul(pa[(int64_t)c * plane], pb[(int64_t)c * plane]));
        }
    const int k = 2 * kr + 1;
    out[gid] = __float2half_rn(acc / (float)(k * k * channel));
}

// backward, stride1 == 1 (correlation_cuda_kernel.cu:151-334).  One thread per input
// element; the reference's reduction order (32 partial sums over tc = l, l+32, ...,
// then a sequential sum of the partials) is kept.
template <bool SECOND>
__global__ __launch_bounds__(256) void corr_backward(
    const float* __restrict__ other, const float* __restrict__ gout, float* __restrict__ gin,
    int batch, int channel, int h, int w, int oc, int oh, int ow,
    int pad, int kr, int md, int s2, int dr) {
    const int64_t total = (int64_t)batch * channel * h * w;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const int bx = (int)(gid % w);
    const int by = (int)((gid / w) % h);
    const int c = (int)((gid / ((int64_t)w * h)) % channel);
    const int n = (int)(gid / ((int64_t)w * h * channel));
    const int dsz = 2 * dr + 1;
    const int y = by + pad, x = bx + pad;                   // padded coordinates
    const float* of = other + ((int64_t)n * channel + c) * h * w;
    const float* go = gout + (int64_t)n * oc * oh * ow;
    const float nelems = (float)((2 * kr + 1) * (2 * kr + 1) * channel);
    float r = 0.0f;
    bool any = SECOND;
    if constexpr (!SECOND) {
        const int xmin = x - kr - md, ymin = y - kr - md, xmax = x + kr - md, ymax = y + kr - md;
        any = !(xmax < 0 || ymax < 0 || xmin >= ow || ymin >= oh || xmin > xmax || ymin > ymax);
    }
    if (any) {
        for (int l = 0; l < 32; ++l) {
            float s = 0.0f;
            for (int tc = l; tc < oc; tc += 32) {
                const int i2 = (tc % dsz - dr) * s2, j2 = (tc / dsz - dr) * s2;
                int xmin, ymin, xmax, ymax;
                float val;
                if constexpr (SECOND) {
                    xmin = x - kr - md - i2; ymin = y - kr - md - j2;
                    xmax = x + kr - md - i2; ymax = y + kr - md - j2;
                    if (xmax < 0 || ymax < 0 || xmin >= ow || ymin >= oh || xmin > xmax || ymin > ymax) continue;
                    val = padded_at(of, h, w, y - j2 - pad, x - i2 - pad);
                } else {
                    xmin = x - kr - md; ymin = y - kr - md; xmax = x + kr - md; ymax = y + kr - md;
                    val = padded_at(of, h, w, y + j2 - pad, x + i2 - pad);
                }
                xmin = max(0, xmin); xmax = min(ow - 1, xmax);
                ymin = max(0, ymin); ymax = min(oh - 1, ymax);
                const float* g = go + (int64_t)tc * oh * ow;
                for (int j = ymin; j <= ymax; ++j)
                    for (int i = xmin; i <= xmax; ++i) s = fmaf(g[(int64_t)j * ow + i], val, s);
            }
            r += s;
        }
        gin[gid] = r / nelems;
    } else {
        gin[gid] = 0.0f;        // the binding zero-fills gradInput (correlation_cuda.cc:112-113)
    }
}

// Backward for PWC-Net's configuration (k == 1, strides 1, pad == md == 4; round 3).  corr_backward above is one thread per
// gradient element: 81 loads of gradOutput + 81 of the other map + 81 multiply-adds each, and a gradOutput element is
// fetched again by every channel (1.1 ms for both gradients at 32 x 288 x 496).  Here a thread owns one PIXEL: its 81
// gradOutput values (gradInput1: the 81 planes at the pixel; gradInput2: plane tc at the pixel moved back by displacement
// tc) live in registers for all channels, and per channel the workgroup (64x4 pixels) stages the other map's window
// (tile + 4 halo, zero padded) in LDS, double-buffered.  The sum runs in the reference's order -- 32 partial sums over
// tc = l, l + 32, l + 64, added in sequence (correlation_cuda_kernel.cu:162-239, 255-332) -- so the result is corr_backward's
// and the oracle's bit for bit.
template <bool SECOND>
__global__ __launch_bounds__(256) void corr_backward_k1(
    const float* __restrict__ other, const float* __restrict__ gout, float* __restrict__ gin,
    int channel, int h, int w, int groups, int ch_per_group) {
    constexpr int MD = 4, D = 2 * MD + 1, OC = D * D, TW = 64, TH = 4, LW = TW + 2 * MD, LH = TH + 2 * MD;
    __shared__ float win[2][LH][LW];
    const int tid = threadIdx.x, px = tid & 63, py = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int bx = x0 + px, by = y0 + py;
    const int n = blockIdx.z / groups, cg = blockIdx.z - n * groups;
    const int c_begin = cg * ch_per_group, c_end = min(channel, c_begin + ch_per_group);
    const bool in = bx < w && by < h;
    const int64_t plane = (int64_t)h * w;
    const float* go = gout + (int64_t)n * OC * plane;

    // the pixel's 81 gradOutput values (zero where the reference skips the term: gradInput2 near the frame's border)
    float g[OC];
#pragma unroll
    for (int tc = 0; tc < OC; ++tc) {
        const int gy = SECOND ? by - (tc / D - MD) : by, gx = SECOND ? bx - (tc % D - MD) : bx;
        g[tc] = (in && gy >= 0 && gy < h && gx >= 0 && gx < w) ? go[(int64_t)tc * plane + (int64_t)gy * w + gx] : 0.0f;
    }
    // (a term the reference skips -- gradInput2, displaced position outside the frame -- reads the other map at that same
    //  position: the zero padding of the staged window, so the skipped term is fma(0, 0, s) = s)

    auto stage = [&](int c, int buf) {
        const float* of = other + ((int64_t)n * channel + c) * plane;
        for (int e = tid; e < LH * LW; e += 256) {
            const int r = e / LW, col = e - r * LW;
            const int gy = y0 - MD + r, gx = x0 - MD + col;
            win[buf][r][col] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? of[(int64_t)gy * w + gx] : 0.0f;
        }
    };
    if (c_begin >= c_end) return;
    stage(c_begin, 0);
    const float nelems = (float)channel;
    for (int c = c_begin; c < c_end; ++c) {
        const int buf = (c - c_begin) & 1;
        __syncthreads();                                    // window c has been written; window c - 1 has been read
        if (c + 1 < c_end) stage(c + 1, buf ^ 1);
        float r = 0.0f;
#pragma unroll
        for (int l = 0; l < 32; ++l) {
            float s = 0.0f;
#pragma unroll
            for (int tc = l; tc < OC; tc += 32) {
                const int tj = tc / D, ti = tc % D;
                const float v = SECOND ? win[buf][py + 2 * MD - tj][px + 2 * MD - ti] : win[buf][py + tj][px + ti];
                s = fmaf(g[tc], v, s);
            }
            r += s;
        }
        if (in) gin[((int64_t)n * channel + c) * plane + (int64_t)by * w + bx] = r / nelems;
    }
}

}  // namespace vfi

using namespace vfi;

// kernel selection thresholds (development knobs, see vfi_common.h): number of 32x8 tiles from which the
// one-lane-per-pixel kernel is used ...
VFI_KNOB(long long, g_corr_big_threshold, 256);
// ... and number of 16x4 tiles below which the one-thread-per-output kernel is used
VFI_KNOB(long long, g_corr_flat_threshold, 64);     // measured at 1080p: 36 tiles 12 us flat vs 29 us tiled; 144 tiles 40 vs 24
VFI_KNOB(int, g_corr_rows2, 1);                     // two pixels per lane in the tiled kernel (development: 2 = four pixels per lane on every aligned level)
// number of 64x4 tiles from which the four-pixels-per-lane kernel (three displacement rows per workgroup) is used
#ifndef CORR_QUAD_THRESHOLD
#define CORR_QUAD_THRESHOLD 128                      // (1080p pyramid: 64 x 144 x 248 -- 144 tiles -- 19.4 us against 21.4 with two pixels per lane; 96 x 72 x 124 -- 36 tiles -- 23.6 against 17.3)
#endif
VFI_KNOB(long long, g_corr_quad_threshold, CORR_QUAD_THRESHOLD);
#ifdef VFI_DEV
static int g_corr_mfma = 0;                         // the matrix-core kernel for the aligned levels (corr_forward_k1_mfma)
#endif
#ifdef VFI_DEV
extern "C" void vfi_dev_correlation(long long big_threshold, long long flat_threshold, int rows2) {
    g_corr_big_threshold = big_threshold; g_corr_flat_threshold = flat_threshold; g_corr_rows2 = rows2;
}
extern "C" void vfi_dev_correlation_mfma(int on) { g_corr_mfma = on; }
#endif

extern "C" int vfi_correlation_output_dims(int h, int w, int pad_size, int kernel_size, int max_displacement,
                                            int stride1, int stride2, int* out_channels, int* out_h, int* out_w) {
    if (kernel_size <= 0 || stride1 <= 0 || stride2 <= 0 || max_displacement < 0 || pad_size < 0) return VFI_ERR_SHAPE;
    const int kr = (kernel_size - 1) / 2;
    const int border = kr + max_displacement;
    const int dr = max_displacement / stride2;
    if (out_channels) *out_channels = (dr * 2 + 1) * (dr * 2 + 1);
    // ceil(float / float), as correlation_cuda.cc:31-32 computes it
    if (out_h) *out_h = (int)ceilf((float)(h + 2 * pad_size - 2 * border) / (float)stride1);
    if (out_w) *out_w = (int)ceilf((float)(w + 2 * pad_size - 2 * border) / (float)stride1);
    return VFI_OK;
}

// one call, or two calls of equal shape (nitems == 2) in one launch
static int correlation_forward_items(const float* const* in1s, const float* const* in2s, float* const* outs, int nitems, int per,
                                     int channel, int h, int w, int pad_size, int kernel_size, int max_displacement,
                                     int stride1, int stride2, vfi_stream_t stream) {
    int oc, oh, ow;
    if (nitems < 1 || nitems > 2 || per <= 0 || channel <= 0 || h <= 0 || w <= 0) return VFI_ERR_SHAPE;
    for (int i = 0; i < nitems; ++i)
        if (!in1s[i] || !in2s[i] || !outs[i]) return VFI_ERR_SHAPE;
    if (nitems == 2 && outs[0] == outs[1]) return VFI_ERR_SHAPE;
    if (vfi_correlation_output_dims(h, w, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow))
        return VFI_ERR_SHAPE;
    if (oh <= 0 || ow <= 0) return VFI_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int batch = nitems * per;                          // images of the launch
    CorrItems items;
    uintptr_t in_bits = 0, out_bits = 0;
    for (int i = 0; i < 2; ++i) {
        const int k = i < nitems ? i : 0;
        items.in1[i] = in1s[k]; items.in2[i] = in2s[k]; items.out[i] = outs[k];
        in_bits |= reinterpret_cast<uintptr_t>(in1s[k]) | reinterpret_cast<uintptr_t>(in2s[k]);
        out_bits |= reinterpret_cast<uintptr_t>(outs[k]);
    }
    items.per = per;
    const int kr = (kernel_size - 1) / 2, dr = max_displacement / stride2;
    if (kernel_size == 1 && stride1 == 1 && stride2 == 1 && max_displacement == 4) {
        const int64_t big_tiles = (int64_t)((ow + 31) / 32) * ((oh + 7) / 8) * batch;
        const int64_t small_tiles = (int64_t)((ow + 15) / 16) * ((oh + 3) / 4) * batch;
        // 16-byte staging needs rows, planes and bases aligned (plane = h * w floats); the tiled kernel writes a lane's two
        // pixels as one 8-byte store: an output view at an odd element offset of its storage takes the other kernels
        const bool aligned = (w & 3) == 0 && ((max_displacement - pad_size) & 3) == 0 && (in_bits & 15) == 0 && (out_bits & 7) == 0 &&
                             (int64_t)h * w * 4 * (CORR_CC_ROWS + 1) < INT_MAX;      // (a chunk of planes through one buffer descriptor)
        if (small_tiles < g_corr_flat_threshold * nitems) {
            const int64_t total = (int64_t)batch * oc * oh * ow;
            hipLaunchKernelGGL(corr_forward_k1_flat<4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, items,
                               batch, channel, h, w, oh, ow, max_displacement - pad_size);
#ifdef VFI_DEV
        } else if (g_corr_mfma && aligned) {
            const dim3 grid((ow + 63) / 64, (oh + 3) / 4, batch);
            hipLaunchKernelGGL(corr_forward_k1_mfma<4>, grid, dim3(1024, 1, 1), 0, st, items,
                               channel, h, w, oh, ow, max_displacement - pad_size);
#endif
        } else if (g_corr_rows2 && aligned && (out_bits & 15) == 0 &&
                   (g_corr_rows2 == 2 || (int64_t)((ow + 63) / 64) * ((oh + 3) / 4) * batch >= g_corr_quad_threshold * nitems)) {
            const int tiles_x = (ow + 63) / 64, tiles_y = (oh + 3) / 4;
            const int64_t nt = (int64_t)tiles_x * tiles_y * batch;
            if (nt > (1 << 26)) return VFI_ERR_SHAPE;
            const unsigned wgs = (unsigned)((nt + 7) / 8) * 8 * 3;                 // whole groups of 8 XCDs x 3 displacement-row groups
            hipLaunchKernelGGL(corr_forward_k1_quad<4>, dim3(wgs), dim3(64, 3, 1), 0, st, items,
                               channel, h, w, oh, ow, max_displacement - pad_size, tiles_x, tiles_y, (int)nt);
        } else if (g_corr_rows2 && aligned) {
            const dim3 grid((ow + 31) / 32, (oh + 3) / 4, batch);
            hipLaunchKernelGGL(corr_forward_k1_rows2<4>, grid, dim3(64, 9, 1), 0, st, items,
                               channel, h, w, oh, ow, max_displacement - pad_size);
        } else if (big_tiles >= g_corr_big_threshold * nitems) {
            const dim3 grid((ow + 31) / 32, (oh + 7) / 8, batch);
            hipLaunchKernelGGL((corr_forward_k1<4, 32, 8>), grid, dim3(32, 8, 1), 0, st, items,
                               channel, h, w, oh, ow, max_displacement - pad_size);
        } else {
            const dim3 grid((ow + 15) / 16, (oh + 3) / 4, batch);
            hipLaunchKernelGGL(corr_forward_k1_rows<4>, grid, dim3(64, 9, 1), 0, st, items,
                               channel, h, w, oh, ow, max_displacement - pad_size);
        }
        return launch_status();
    }
    // any other configuration: the generic kernel, one launch per item
    for (int i = 0; i < nitems; ++i) {
        const int64_t total = (int64_t)per * oc * oh * ow;
        hipLaunchKernelGGL(corr_forward_generic, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in1s[i],
                           in2s[i], outs[i], per, channel, h, w, oc, oh, ow, pad_size, kr, max_displacement, stride1,
                           stride2, dr);
        if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    }
    return VFI_OK;
}

extern "C" int vfi_correlation_forward(const float* input1, const float* input2, float* output, int batch, int channel,
                                        int h, int w, int pad_size, int kernel_size, int max_displacement,
                                        int stride1, int stride2, vfi_stream_t stream) {
    if (batch <= 0 || !input1 || !input2 || !output) return VFI_ERR_SHAPE;
    return correlation_forward_items(&input1, &input2, &output, 1, batch, channel, h, w, pad_size, kernel_size, max_displacement,
                                     stride1, stride2, stream);
}

// Two correlation calls of equal shape in ONE launch -- the same pyramid level of the two flow networks PWCDCNet runs for a
// frame pair, (I0, I1) and (I1, I0) (networks/DAIN.py:196-202, PWCNet/PWCNet.py:230-300).  Results: the two single calls',
// bit for bit (the same kernels; which of them runs is decided on the pair's tile count, and every one of them sums the
// channels in the same order).
extern "C" int vfi_correlation_forward_pair(const float* input1_a, const float* input2_a, float* output_a,
                                             const float* input1_b, const float* input2_b, float* output_b,
                                             int batch, int channel, int h, int w, int pad_size, int kernel_size,
                                             int max_displacement, int stride1, int stride2, vfi_stream_t stream) {
    if (batch <= 0) return VFI_ERR_SHAPE;
    const float* in1s[2] = {input1_a, input1_b};
    const float* in2s[2] = {input2_a, input2_b};
    float* outs[2] = {output_a, output_b};
    return correlation_forward_items(in1s, in2s, outs, 2, batch, channel, h, w, pad_size, kernel_size, max_displacement,
                                     stride1, stride2, stream);
}

extern "C" int vfi_correlation_forward_f16(const void* input1, const void* input2, void* output, int batch, int channel,
                                            int h, int w, int pad_size, int kernel_size, int max_displacement,
                                            int stride1, int stride2, vfi_stream_t stream) {
    int oc, oh, ow;
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !output) return VFI_ERR_SHAPE;
    if (vfi_correlation_output_dims(h, w, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow))
        return VFI_ERR_SHAPE;
    if (oh <= 0 || ow <= 0) return VFI_ERR_SHAPE;
    const int kr = (kernel_size - 1) / 2, dr = max_displacement / stride2;
    const int64_t total = (int64_t)batch * oc * oh * ow;
    if ((total + 255) / 256 > INT_MAX) return VFI_ERR_SHAPE;
    if (kernel_size == 1 && stride1 == 1 && stride2 == 1 && max_displacement == 4) {
        // PWC-Net's configuration on frames whose rows are 8-byte aligned, enough tiles to fill the chip: the tiled kernel
        const int64_t small_tiles = (int64_t)((ow + 15) / 16) * ((oh + 3) / 4) * batch;
        const bool aligned = (w & 3) == 0 && ((max_displacement - pad_size) & 3) == 0 &&
                             ((reinterpret_cast<uintptr_t>(input1) | reinterpret_cast<uintptr_t>(input2)) & 7) == 0 &&
                             (int64_t)h * w * 2 * (CORR_CC_ROWS + 1) < INT_MAX;      // (a chunk of planes through one buffer descriptor)
        // (measured at the 1080p pyramid: 144 such tiles 27 us tiled vs 21 us one thread per output; 576 tiles 33 vs 101)
        if (aligned && small_tiles >= 4 * g_corr_flat_threshold) {
            const dim3 grid((ow + 31) / 32, (oh + 3) / 4, batch);
            hipLaunchKernelGGL(corr_forward_k1_rows2_f16<4>, grid, dim3(64, 9, 1), 0, (hipStream_t)stream, (const __half*)input1,
                               (const __half*)input2, (__half*)output, channel, h, w, oh, ow, max_displacement - pad_size);
            return launch_status();
        }
    }
    hipLaunchKernelGGL(corr_forward_generic_f16, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __half*)input1, (const __half*)input2, (__half*)output, batch, channel, h, w, oc, oh, ow,
                       pad_size, kr, max_displacement, stride1, stride2, dr);
    return launch_status();
}

extern "C" int vfi_correlation_backward(const float* input1, const float* input2, const float* gradoutput,
                                         float* gradinput1, float* gradinput2, int batch, int channel, int h, int w,
                                         int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                         vfi_stream_t stream) {
    int oc, oh, ow;
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0) return VFI_ERR_SHAPE;
    if (!input1 || !input2 || !gradoutput || !gradinput1 || !gradinput2) return VFI_ERR_SHAPE;
    if (vfi_correlation_output_dims(h, w, pad_size, kernel_size, max_displacement, stride1, stride2, &oc, &oh, &ow))
        return VFI_ERR_SHAPE;
    // the reference's backward indexes gradInput rows by blockIdx*stride1 and leaves
    // the tensor for stride1 > 1: only stride1 == 1 is defined
    if (stride1 != 1 || oh <= 0 || ow <= 0) return VFI_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int kr = (kernel_size - 1) / 2, dr = max_displacement / stride2;
    if (kernel_size == 1 && stride2 == 1 && max_displacement == 4 && pad_size == 4) {
        // PWC-Net's configuration: the pixel-owns-its-gradOutput kernel; channel groups over blockIdx.z fill the chip
        const int tiles = ((w + 63) / 64) * ((h + 3) / 4);
        int groups = (int)std::min<int64_t>(channel, std::max<int64_t>(1, (2048 + (int64_t)tiles * batch - 1) / ((int64_t)tiles * batch)));
        const int ch_per_group = (channel + groups - 1) / groups;
        groups = (channel + ch_per_group - 1) / ch_per_group;
        if ((int64_t)batch * groups <= 65535) {
            const dim3 grid((w + 63) / 64, (h + 3) / 4, batch * groups);
            hipLaunchKernelGGL(corr_backward_k1<false>, grid, dim3(256), 0, st, input2, gradoutput, gradinput1, channel, h, w, groups, ch_per_group);
            if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
            hipLaunchKernelGGL(corr_backward_k1<true>, grid, dim3(256), 0, st, input1, gradoutput, gradinput2, channel, h, w, groups, ch_per_group);
            return launch_status();
        }
    }
    const int64_t total = (int64_t)batch * channel * h * w;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipLaunchKernelGGL(corr_backward<false>, grid, block, 0, st, input2, gradoutput, gradinput1, batch, channel, h, w,
                       oc, oh, ow, pad_size, kr, max_displacement, stride2, dr);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    hipLaunchKernelGGL(corr_backward<true>, grid, block, 0, st, input1, gradoutput, gradinput2, batch, channel, h, w,
                       oc, oh, ow, pad_size, kr, max_displacement, stride2, dr);
    return launch_status();
}
