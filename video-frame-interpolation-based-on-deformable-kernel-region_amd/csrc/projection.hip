// projection.hip -- FlowProjection / DepthFlowProjection (forward splat of the
// t->0 flow to the intermediate time) for gfx950.
//
// Semantics: flowprojection_cuda_kernel.cu:29-301 and
// depthflowprojection_cuda_kernel.cu:29-341 of the reference; entry points
// replace flowprojection_cuda.cc / depthflowprojection_cuda.cc.
//
// What is computed (three steps in the reference, each its own launch there):
//   1. splat: every source pixel adds (-d*fx, -d*fy, d) to its 4 integer
//      neighbours (d = 1 without depth) -- 12 global atomics per pixel there;
//   2. normalise where count > 0;
//   3. optional hole fill from the nearest non-hole in -x, +x, -y, +y (needs the
//      complete count plane, so it stays a separate launch).
//
// How (owner computes, pull; no global atomics on the normal path).  The frame is cut into 64x16
// OUTPUT tiles, into 16x16 source BLOCKS and into 256x64 SUPER-tiles of 64 blocks.
//   K0  proj_scan    one workgroup per super-tile: per block the range of integer displacements
//                    (min/max of L - x, R - x, T - y, B - y over its valid pixels) and the largest
//                    addends, per super-tile the box of all its targets.  Reads the flow once.
//   K1  proj_pull    one workgroup per output tile: finds the super-tiles whose target box meets the
//                    tile, among their blocks those that can reach it, clipped to the pixels that can;
//                    walks the bounding rectangle U of those pixels (for a smooth field U is the tile
//                    shifted by the flow and a few pixels larger: ~1.3 source pixels per output pixel),
//                    accumulates in LDS and writes count and the normalised flow once, coalesced,
//                    plus two bitmaps of "count != 0" and the tile's number of holes.
//   K2  proj_finish  hole filling for the tiles that have holes (bitmap walks instead of the
//                    reference's cell-by-cell walks); resets the per-call state.
// K1 writes every cell of count and output, so callers need not zero-fill them (the reference's
// callers must: its splat accumulates into them).  Nothing of a call's state crosses to the host:
// the three launches are fixed, which makes a captured graph replayable.
//
// Fallback: K0 also sums how many (block, output tile) pairs there are; for fields whose blocks
// reach many tiles each (random flow of +-W/2) K1 instead splats its own tile with global atomics
// exactly like the reference -- into three scratch planes of the workspace that are zero between
// calls -- and K2 normalises them into count / output and fills holes from them.  (The scratch planes
// are cleaned by the next call's K0, whose workgroups see a "dirty" word.)
//
// Accumulation in K1 is fixed point in LDS integer atomics: on gfx950 an LDS float atomic add costs
// ~170 cycles per wave instruction (tools/probes/lds_atomic_probe.hip), an integer one a few.  Every
// addend is scaled by a power of two chosen per output tile from the largest |addend| that can reach it
// (from K0's block table) so that it is below 2^25, rounded to an integer, and the two flow components
// are added as ONE 64-bit integer (x << 32) + y: with at most 32 addends per cell neither half leaves
// its 32 bits, and the halves are separated exactly afterwards.  The number of addends per cell is
// accumulated beside it (it IS the count plane of FlowProjection); a tile with a busier cell (flows
// converging 8-fold) is accumulated again with a correspondingly coarser scale.  Sums are exact integers,
// so the result does not depend on the summation order -- reproducible bit for bit from run to run
// (the reference's fp32 atomic sum carries one rounding per addend, in arrival order) -- and agrees
// with any fp32 summation order to rounding; addends that are multiples of 2^-k (k < ~16) sum exactly
// in both.  count of FlowProjection is exact.
#include "vfi_common.h"
#include "bitwalk.h"
#include "workspace.h"

#include <limits.h>

namespace vfi {

#define PROJ_TW 64                  // output tile
#define PROJ_TH 16
#define PROJ_THREADS 256
#define PROJ_BLK 16                 // source block edge
#define PROJ_SUP_W 256              // super-tile = 16 x 4 blocks, one K0 workgroup
#define PROJ_SUP_H 64
#define PROJ_SUP_BLOCKS 64
#define PROJ_SCAN_THREADS 1024
#define PROJ_MAXHIT 64              // super-tiles listed per output tile before K1 scans all of them
#define PROJ_ADD_BITS 25            // |scaled addend| < 2^25
#define PROJ_ADD_CELL 32            // addends per cell that fit beside it in 32 bits
#define PROJ_COST_LIMIT 48          // (block, tile) pairs per block, frame average, before the fallback

// workspace "words" (32-bit).  Header: [0..1] 64-bit number of (block, output tile) pairs of this call
// (summed by K0, read by K1, reset by K2); [2] the scratch planes of the fallback hold sums (written by
// K1, read by K2 and by the next call's K0).  Then one int4 per super-tile (target box x0, y0, x1, y1;
// x0 > x1: none), one int4 per block (dxmin | dxmax << 16, dymin | dymax << 16, bits of the largest
// |value addend|, bits of the largest |count addend|; min > max: no valid pixel) -- block j of super-tile
// s at index 64 s + j, so a wave reads a super-tile's blocks with one load -- and one word per output tile
// (its number of holes).  All of it is rewritten by every call; only the header carries state.
// workspace "bits": two bitmaps of "count != 0", one packed along rows (rowmap[b][y][x/32]) and one packed
// along columns (colmap[b][x][y/32]), written by K1 for the hole filler.
#define PROJ_WS_HDR 16
#define PROJ_WS_COST 0
#define PROJ_WS_DIRTY 2

// rmw / cmw: 32-bit words per image row / column of the two bitmaps; rowmap / colmap: their word offsets
// inside the bit buffer; sup_x, sup_y, nsup: super-tiles per row / column / image; off_*: word offsets of the
// tables inside the word buffer
struct ProjGeom {
    int h, w, tiles_x, tiles_y, ntiles, rmw, cmw, rowmap, colmap;
    int sup_x, sup_y, nsup, off_sup, off_blk, off_holes;
};

// one source pixel: validity, the four target cells (in order TL, TR, BL, BR) and the three addends
struct ProjSplat {
    bool valid;
    int L, T, R, Bm;
    float ax, ay, ac;
};

// Where the flow of a source pixel comes from.  UP == false: the full-resolution flow tensor of the
// reference's FlowProjection.  UP == true: the network's quarter-resolution flow; the pixel's flow is
// nn.Upsample(scale_factor=4, mode='bilinear') of (m0 * flow) * m1, formed on the fly -- the x4
// upsampled tensor of forward_flownets (networks/DAIN_slowmotion.py:204-216) is never materialised.
struct ProjFlow {
    const float* p;
    vfi_strides s;
    int hq, wq;             // quarter-resolution size (UP only)
    float m0, m1;           // div_flow, time offset (UP only)
};

// torch's upsample_bilinear2d, align_corners=False, scale factor 4 (ATen UpSampleBilinear2d):
// source index 0.25 * (dst + 0.5) - 0.5 clamped at 0, second tap one further unless at the edge
struct UpTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ UpTap up4_tap(int dst, int in_size) {
    float src = 0.25f * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    UpTap t;
    t.i0 = (int)src;
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.0f - t.l1;
    return t;
}
// one channel of the upsampled (m0 * plane) * m1 from its four taps; fused as nvcc fuses ATen's expression
__device__ __forceinline__ float up4_blend(float q00, float q01, float q10, float q11, const UpTap& ty, const UpTap& tx,
                                           float m0, float m1) {
    const float p00 = (m0 * q00) * m1, p01 = (m0 * q01) * m1, p10 = (m0 * q10) * m1, p11 = (m0 * q11) * m1;
    const float t0 = fmaf(tx.l1, p01, tx.l0 * p00);
    const float t1 = fmaf(tx.l1, p11, tx.l0 * p10);
    return fmaf(ty.l1, t1, ty.l0 * t0);
}
__device__ __forceinline__ float up4_sample(const float* __restrict__ plane, int64_t hs, const UpTap& ty, const UpTap& tx,
                                            float m0, float m1) {
    return up4_blend(plane[(int64_t)ty.i0 * hs + tx.i0], plane[(int64_t)ty.i0 * hs + tx.i1],
                     plane[(int64_t)ty.i1 * hs + tx.i0], plane[(int64_t)ty.i1 * hs + tx.i1], ty, tx, m0, m1);
}

// A source pixel in two steps, so that a caller can have the loads of the next pixel in flight while
// it works on the current one: proj_load only issues loads (raw values, no arithmetic on them),
// proj_make turns them into the splat.
struct ProjRaw {
    bool in;                // inside the frame
    int x, y;
    float v[8];             // !UP: v[0] = fx, v[1] = fy;  UP: the 2 x 4 quarter-resolution taps
    float d;                // depth weight (DEPTH only)
};

template <bool DEPTH, bool UP>
__device__ __forceinline__ ProjRaw proj_load(const ProjFlow& f, const float* __restrict__ in2,
                                             int b, int x, int y, int h, int w, vfi_strides s2) {
    ProjRaw r;
    r.in = x < w && y < h;
    r.x = x; r.y = y;
#pragma unroll
    for (int k = 0; k < 8; ++k) r.v[k] = 0.0f;
    r.d = 0.0f;
    if (!r.in) return r;
    if constexpr (UP) {
        const UpTap ty = up4_tap(y, f.hq), tx = up4_tap(x, f.wq);
        const float* q = f.p + (int64_t)b * f.s.b;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* plane = q + (int64_t)c * f.s.c;
            r.v[4 * c + 0] = plane[(int64_t)ty.i0 * f.s.h + tx.i0]; r.v[4 * c + 1] = plane[(int64_t)ty.i0 * f.s.h + tx.i1];
            r.v[4 * c + 2] = plane[(int64_t)ty.i1 * f.s.h + tx.i0]; r.v[4 * c + 3] = plane[(int64_t)ty.i1 * f.s.h + tx.i1];
        }
    } else {
        const float* flow = f.p + (int64_t)b * f.s.b + (int64_t)y * f.s.h + x;
        r.v[0] = flow[0];
        r.v[1] = flow[f.s.c];
    }
    if constexpr (DEPTH) r.d = in2[(int64_t)b * s2.b + (int64_t)y * s2.h + x];
    return r;
}

template <bool DEPTH, bool UP>
__device__ __forceinline__ ProjSplat proj_make(const ProjFlow& f, const ProjRaw& r, int h, int w) {
    ProjSplat s;
    s.valid = false;
    s.L = s.T = s.R = s.Bm = 0;
    s.ax = s.ay = s.ac = 0.0f;
    if (!r.in) return s;
    float fx, fy;
    if constexpr (UP) {
        const UpTap ty = up4_tap(r.y, f.hq), tx = up4_tap(r.x, f.wq);
        fx = up4_blend(r.v[0], r.v[1], r.v[2], r.v[3], ty, tx, f.m0, f.m1);
        fy = up4_blend(r.v[4], r.v[5], r.v[6], r.v[7], ty, tx, f.m0, f.m1);
    } else {
        fx = r.v[0];
        fy = r.v[1];
    }
    const float x2 = (float)r.x + fx;
    const float y2 = (float)r.y + fy;
    if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(w - 1) && y2 <= (float)(h - 1))) return s;
    s.valid = true;
    s.L = (int)x2;
    s.T = (int)y2;
    s.R = min(s.L + 1, w - 1);
    s.Bm = min(s.T + 1, h - 1);
    if constexpr (DEPTH) {
        s.ax = -r.d * fx; s.ay = -r.d * fy; s.ac = r.d;     // depthflowprojection_cuda_kernel.cu:74-91
    } else {
        s.ax = -fx; s.ay = -fy; s.ac = 1.0f;                // flowprojection_cuda_kernel.cu:75-88
    }
    return s;
}

template <bool DEPTH, bool UP>
__device__ __forceinline__ ProjSplat proj_source(const ProjFlow& f, const float* __restrict__ in2,
                                                 int b, int x, int y, int h, int w, vfi_strides s2) {
    return proj_make<DEPTH, UP>(f, proj_load<DEPTH, UP>(f, in2, b, x, y, h, w, s2), h, w);
}


// min / max over each row of 16 lanes (four DPP row_shr steps); the result is in lane 15 of the row
#define PROJ_ROW_STEP(OP, CTRL) v = OP(v, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false))
__device__ __forceinline__ int row16_min(int v) {
    PROJ_ROW_STEP(min, 0x111); PROJ_ROW_STEP(min, 0x112); PROJ_ROW_STEP(min, 0x114); PROJ_ROW_STEP(min, 0x118);
    return v;
}
__device__ __forceinline__ int row16_max(int v) {
    PROJ_ROW_STEP(max, 0x111); PROJ_ROW_STEP(max, 0x112); PROJ_ROW_STEP(max, 0x114); PROJ_ROW_STEP(max, 0x118);
    return v;
}
#undef PROJ_ROW_STEP

__device__ __forceinline__ int pack16(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int lo16(int v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int hi16(int v) { return v >> 16; }

// K0: displacement ranges per 16x16 block, target box per super-tile.  A wave covers 64 x 16 pixels (four
// blocks side by side: lane = x, 16 rows in a loop), the 16 waves of a workgroup 4 x 4 of those.
template <bool DEPTH, bool UP>
__global__ __launch_bounds__(PROJ_SCAN_THREADS) void proj_scan(
    ProjFlow flow, const float* __restrict__ in2, ProjGeom g, vfi_strides s2,
    int* __restrict__ ws, float* __restrict__ planes, int64_t plane_floats) {
    if (ws[PROJ_WS_DIRTY] != 0) {
        // the previous call on this workspace took the fallback: its scratch planes are cleaned here,
        // a slice per workgroup (this call's K1 rewrites the word)
        const int64_t chunk = (plane_floats + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(plane_floats, lo + chunk);
        for (int64_t i = lo + threadIdx.x; i < hi; i += PROJ_SCAN_THREADS) planes[i] = 0.0f;
    }
    __shared__ int sbox[4];
    __shared__ int scost;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sup = blockIdx.x;
    const int b = sup / g.nsup;
    const int srem = sup - b * g.nsup;
    const int sy = srem / g.sup_x, sx = srem - sy * g.sup_x;
    if (tid < 4) sbox[tid] = tid < 2 ? INT_MAX : INT_MIN;
    if (tid == 4) scost = 0;
    __syncthreads();
    const int x = sx * PROJ_SUP_W + (wave & 3) * 64 + lane;
    const int y0 = sy * PROJ_SUP_H + (wave >> 2) * PROJ_BLK;
    int dxmin = INT_MAX, dxmax = INT_MIN, dymin = INT_MAX, dymax = INT_MIN, vbits = 0, cbits = 0;
#pragma unroll 1
    for (int r0 = 0; r0 < PROJ_BLK; r0 += 4) {
        ProjRaw raw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) raw[k] = proj_load<DEPTH, UP>(flow, in2, b, x, y0 + r0 + k, g.h, g.w, s2);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const ProjSplat s = proj_make<DEPTH, UP>(flow, raw[k], g.h, g.w);
            if (s.valid) {
                dxmin = min(dxmin, s.L - x); dxmax = max(dxmax, s.R - x);
                dymin = min(dymin, s.T - raw[k].y); dymax = max(dymax, s.Bm - raw[k].y);
                // non-negative floats order like their bit patterns
                vbits = max(vbits, __float_as_int(fmaxf(fabsf(s.ax), fabsf(s.ay))));
                cbits = max(cbits, __float_as_int(fabsf(s.ac)));
            }
        }
    }
    dxmin = row16_min(dxmin); dxmax = row16_max(dxmax);
    dymin = row16_min(dymin); dymax = row16_max(dymax);
    vbits = row16_max(vbits); cbits = row16_max(cbits);
    if ((lane & 15) == 15) {
        const bool any = dxmin != INT_MAX;
        int4 e;
        e.x = any ? pack16(dxmin, dxmax) : pack16(1, 0);
        e.y = any ? pack16(dymin, dymax) : pack16(1, 0);
        e.z = vbits;
        e.w = cbits;
        const int idx = (wave >> 2) * 16 + (wave & 3) * 4 + (lane >> 4);
        reinterpret_cast<int4*>(ws + g.off_blk)[(int64_t)sup * PROJ_SUP_BLOCKS + idx] = e;
        if (any) {
            const int bx0 = x - 15, bx1 = min(x, g.w - 1), by1 = min(y0 + PROJ_BLK - 1, g.h - 1);
            const int X0 = max(bx0 + dxmin, 0), X1 = min(bx1 + dxmax, g.w - 1);
            const int Y0 = max(y0 + dymin, 0), Y1 = min(by1 + dymax, g.h - 1);
            atomicMin(&sbox[0], X0); atomicMin(&sbox[1], Y0);
            atomicMax(&sbox[2], X1); atomicMax(&sbox[3], Y1);
            atomicAdd(&scost, (X1 / PROJ_TW - X0 / PROJ_TW + 1) * (Y1 / PROJ_TH - Y0 / PROJ_TH + 1));
        }
    }
    __syncthreads();
    if (tid == 0) {
        reinterpret_cast<int4*>(ws + g.off_sup)[sup] = make_int4(sbox[0], sbox[1], sbox[2], sbox[3]);
        if (scost) atomicAdd(reinterpret_cast<unsigned long long*>(ws + PROJ_WS_COST), (unsigned long long)scost);
    }
}

// the two halves of a packed sum, exactly: S = hi * 2^32 + lo with both in int32
__device__ __forceinline__ unsigned long long pack2(int hi, int lo) {
    return ((unsigned long long)(unsigned)hi << 32) + (unsigned long long)(long long)lo;
}
__device__ __forceinline__ int packed_lo(unsigned long long s) { return (int)(unsigned)s; }
__device__ __forceinline__ int packed_hi(unsigned long long s) {
    return (int)((s - (unsigned long long)(long long)packed_lo(s)) >> 32);
}

template <bool DEPTH> struct ProjCountCell { typedef unsigned type; };
template <> struct ProjCountCell<true> { typedef unsigned long long type; };    // (addends << 32) + scaled weight sum

// K1: one workgroup per output tile
template <bool DEPTH, bool UP>
__global__ __launch_bounds__(PROJ_THREADS, 8) void proj_pull(
    ProjFlow flow, const float* __restrict__ in2, float* __restrict__ count, float* __restrict__ out,
    ProjGeom g, vfi_strides s1, vfi_strides s2, vfi_strides sc, int* __restrict__ ws, int* __restrict__ bits,
    float* __restrict__ planes, unsigned long long cost_limit) {
    typedef typename ProjCountCell<DEPTH>::type ccell;
    __shared__ unsigned long long accv[PROJ_TH][PROJ_TW];
    __shared__ ccell accc[PROJ_TH][PROJ_TW];
    __shared__ int s_hits[PROJ_MAXHIT];
    __shared__ int s_st[8];             // 0 hits, 1..4 box x0 y0 x1 y1, 5 / 6 bits of the largest addends, 7 most addends in a cell
    __shared__ unsigned s_colm[PROJ_TW];
    __shared__ int s_holes;
    const int tile = blockIdx.x;
    const int per_img = g.tiles_x * g.tiles_y;
    const int b = tile / per_img;
    const int trem = tile - b * per_img;
    const int tyi = trem / g.tiles_x, txi = trem - tyi * g.tiles_x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int ox0 = txi * PROJ_TW, oy0 = tyi * PROJ_TH;
    const bool fallback = *reinterpret_cast<const unsigned long long*>(ws + PROJ_WS_COST) > cost_limit;
    if (tile == 0 && tid == 0) ws[PROJ_WS_DIRTY] = fallback ? 1 : 0;

    if (fallback) {
        // the reference's own scheme: this tile as SOURCE tile, global atomics into the dense scratch
        // planes [value x | value y | count][batch][h][w] of the workspace (zero between calls)
        const int64_t npx = (int64_t)(g.ntiles / per_img) * g.h * g.w;
        float* o0 = planes + (int64_t)b * g.h * g.w;
        float* o1 = o0 + npx;
        float* cn = o1 + npx;
#pragma unroll
        for (int r = 0; r < PROJ_TH / 4; ++r) {
            const ProjSplat s = proj_source<DEPTH, UP>(flow, in2, b, ox0 + lane, oy0 + wave + r * 4, g.h, g.w, s2);
            if (!s.valid) continue;
            const int64_t oT = (int64_t)s.T * g.w, oB = (int64_t)s.Bm * g.w;
            atomicAdd(&o0[oT + s.L], s.ax); atomicAdd(&o0[oT + s.R], s.ax); atomicAdd(&o0[oB + s.L], s.ax); atomicAdd(&o0[oB + s.R], s.ax);
            atomicAdd(&o1[oT + s.L], s.ay); atomicAdd(&o1[oT + s.R], s.ay); atomicAdd(&o1[oB + s.L], s.ay); atomicAdd(&o1[oB + s.R], s.ay);
            atomicAdd(&cn[oT + s.L], s.ac); atomicAdd(&cn[oT + s.R], s.ac); atomicAdd(&cn[oB + s.L], s.ac); atomicAdd(&cn[oB + s.R], s.ac);
        }
        return;
    }

    if (tid < 8) s_st[tid] = (tid == 1 || tid == 2) ? INT_MAX : (tid == 3 || tid == 4) ? INT_MIN : 0;
    if (tid < PROJ_TW) s_colm[tid] = 0u;
    if (tid == 8) s_holes = 0;
    for (int i = tid; i < PROJ_TH * PROJ_TW; i += PROJ_THREADS) { (&accv[0][0])[i] = 0ull; (&accc[0][0])[i] = 0; }
    __syncthreads();

    // ---- which pixels can reach this tile
    const int tx1 = min(ox0 + PROJ_TW - 1, g.w - 1), ty1 = min(oy0 + PROJ_TH - 1, g.h - 1);
    const int4* supt = reinterpret_cast<const int4*>(ws + g.off_sup) + (int64_t)b * g.nsup;
    for (int s = tid; s < g.nsup; s += PROJ_THREADS) {
        const int4 e = supt[s];
        if (e.x <= tx1 && e.z >= ox0 && e.y <= ty1 && e.w >= oy0) {
            const int k = atomicAdd(&s_st[0], 1);
            if (k < PROJ_MAXHIT) s_hits[k] = s;
        }
    }
    __syncthreads();
    {
        const int nhit = s_st[0];
        const bool all = nhit > PROJ_MAXHIT;                // (then every super-tile is scanned: a superset)
        const int nscan = all ? g.nsup : nhit;
        const int4* blkt = reinterpret_cast<const int4*>(ws + g.off_blk) + (int64_t)b * g.nsup * PROJ_SUP_BLOCKS;
        for (int k = wave; k < nscan; k += PROJ_THREADS / 64) {
            const int s = all ? k : s_hits[k];
            const int4 e = blkt[(int64_t)s * PROJ_SUP_BLOCKS + lane];
            const int ssy = s / g.sup_x, ssx = s - ssy * g.sup_x;
            const int bx0 = ssx * PROJ_SUP_W + (lane & 15) * PROJ_BLK, by0 = ssy * PROJ_SUP_H + (lane >> 4) * PROJ_BLK;
            const int bx1 = min(bx0 + PROJ_BLK - 1, g.w - 1), by1 = min(by0 + PROJ_BLK - 1, g.h - 1);
            const int dxmin = lo16(e.x), dxmax = hi16(e.x), dymin = lo16(e.y), dymax = hi16(e.y);
            // a pixel at x reaches columns [x + dxmin, x + dxmax] at most
            const int sx0 = max(bx0, ox0 - dxmax), sx1 = min(bx1, tx1 - dxmin);
            const int sy0 = max(by0, oy0 - dymax), sy1 = min(by1, ty1 - dymin);
            if (dxmin <= dxmax && sx0 <= sx1 && sy0 <= sy1) {
                atomicMin(&s_st[1], sx0); atomicMin(&s_st[2], sy0);
                atomicMax(&s_st[3], sx1); atomicMax(&s_st[4], sy1);
                atomicMax(&s_st[5], e.z); atomicMax(&s_st[6], e.w);
            }
        }
    }
    __syncthreads();
    const int ux0 = s_st[1], uy0 = s_st[2];
    const int uw = s_st[3] - ux0 + 1, uh = s_st[3] >= ux0 ? s_st[4] - uy0 + 1 : 0;   // uh == 0: nothing lands here
    // fixed-point scales: every addend that reaches this tile is below 2^e with e from the blocks' maxima, so
    // addend * 2^(25 - e) is below 2^25 in magnitude
    int ev = 0, ec = 0;
    (void)frexpf(__int_as_float(s_st[5]), &ev);
    (void)frexpf(__int_as_float(s_st[6]), &ec);
    // (clamped so that 2^k stays a normal float when every addend is tiny or huge)
    int kv = max(-100, min(100, PROJ_ADD_BITS - ev)), kc = max(-100, min(100, PROJ_ADD_BITS - ec));

    for (int attempt = 0;; ++attempt) {
        const float sv = ldexpf(1.0f, kv), scn = ldexpf(1.0f, kc);     // exact powers of two
        if (uh > 0) {
            // thread i takes pixels i, i + 256, ... of U in row-major order (lanes = consecutive x); the next
            // pixel's loads are in flight while the current one is accumulated
            const int stepx = PROJ_THREADS % uw, stepy = PROJ_THREADS / uw;
            int yi = tid / uw, xi = tid - yi * uw;
            ProjRaw nxt = proj_load<DEPTH, UP>(flow, in2, b, ux0 + xi, yi < uh ? uy0 + yi : g.h, g.h, g.w, s2);
            while (yi < uh) {
                const ProjRaw cur = nxt;
                xi += stepx; yi += stepy;
                if (xi >= uw) { xi -= uw; yi += 1; }
                nxt = proj_load<DEPTH, UP>(flow, in2, b, ux0 + xi, yi < uh ? uy0 + yi : g.h, g.h, g.w, s2);
                const ProjSplat s = proj_make<DEPTH, UP>(flow, cur, g.h, g.w);
                // cells of this output tile only; R == L / Bm == T at the far edges add twice (:72-73)
                const int lx = s.L - ox0, rx = s.R - ox0, ty = s.T - oy0, by = s.Bm - oy0;
                const bool inL = s.valid && (unsigned)lx < PROJ_TW, inR = s.valid && (unsigned)rx < PROJ_TW;
                const bool inT = (unsigned)ty < PROJ_TH, inB = (unsigned)by < PROJ_TH;
                // addend * 2^k is exact in float (power-of-two scale)
                const unsigned long long qv = pack2(__float2int_rn(s.ax * sv), __float2int_rn(s.ay * sv));
                ccell qc;
                if constexpr (DEPTH) qc = pack2(1, __float2int_rn(s.ac * scn)); else qc = 1u;
                if (inT && inL) { atomicAdd(&accv[ty][lx], qv); atomicAdd(&accc[ty][lx], qc); }
                if (inT && inR) { atomicAdd(&accv[ty][rx], qv); atomicAdd(&accc[ty][rx], qc); }
                if (inB && inL) { atomicAdd(&accv[by][lx], qv); atomicAdd(&accc[by][lx], qc); }
                if (inB && inR) { atomicAdd(&accv[by][rx], qv); atomicAdd(&accc[by][rx], qc); }
            }
        }
        __syncthreads();
        if (attempt) break;
        // did every cell stay within the addends its 32-bit halves can hold?
        int nmax = 0;
#pragma unroll
        for (int r = 0; r < PROJ_TH / 4; ++r) {
            const ccell c = accc[wave + r * 4][lane];
            if constexpr (DEPTH) nmax = max(nmax, packed_hi(c)); else nmax = max(nmax, (int)min(c, 0x7fffffffu));
        }
        nmax = wave_max_i32(nmax);
        if (lane == 0 && nmax > PROJ_ADD_CELL) atomicMax(&s_st[7], nmax);
        __syncthreads();
        nmax = s_st[7];
        if (nmax <= PROJ_ADD_CELL) break;
        // once more with addends small enough for the busiest cell
        const int shift = (32 - __clz(nmax - 1)) - 5;           // ceil(log2(nmax)) - log2(32)
        kv -= shift; kc -= shift;
        for (int i = tid; i < PROJ_TH * PROJ_TW; i += PROJ_THREADS) { (&accv[0][0])[i] = 0ull; (&accc[0][0])[i] = 0; }
        __syncthreads();
    }

    // normalise (flowprojection_cuda_kernel.cu:129-134) and write the tile once; leave the two
    // "count != 0" bitmaps and the number of holes for the hole filler
    const int x = ox0 + lane;
    float cv[PROJ_TH / 4], vxv[PROJ_TH / 4], vyv[PROJ_TH / 4];
    unsigned mine = 0u;
    int holes = 0;
#pragma unroll
    for (int r = 0; r < PROJ_TH / 4; ++r) {
        const int yl = wave + r * 4;
        const bool inside = x < g.w && oy0 + yl < g.h;
        // exact integer sums -> float once
        const unsigned long long sv2 = accv[yl][lane];
        const ccell cc = accc[yl][lane];
        float c;
        if constexpr (DEPTH) c = ldexpf((float)packed_lo(cc), -kc); else c = (float)cc;
        float vx = ldexpf((float)packed_hi(sv2), -kv), vy = ldexpf((float)packed_lo(sv2), -kv);
        if (c > 0.0f) { vx /= c; vy /= c; }
        cv[r] = c; vxv[r] = vx; vyv[r] = vy;
        const bool nz = inside && c != 0.0f;
        const unsigned long long rowbits = __ballot(nz);
        if (lane < 2 && oy0 + yl < g.h && txi * 2 + lane < g.rmw)
            bits[g.rowmap + (b * g.h + oy0 + yl) * g.rmw + txi * 2 + lane] = (int)(unsigned)(rowbits >> (32 * lane));
        if (nz) mine |= 1u << yl;
        holes += __popcll(__ballot(inside && c <= 0.0f));
    }
    if (mine) atomicOr(&s_colm[lane], mine);
    if (lane == 0 && holes) atomicAdd(&s_holes, holes);
    __syncthreads();
    // a column word holds two tiles' rows: each tile stores its own 16-bit half
    if (tid < PROJ_TW && ox0 + tid < g.w)
    {
        unsigned short* half = reinterpret_cast<unsigned short*>(bits + g.colmap) + ((int64_t)(b * g.w + ox0 + tid) * g.cmw) * 2;
        half[tyi] = (unsigned short)s_colm[tid];
        if (tyi == g.tiles_y - 1 && (tyi & 1) == 0) half[tyi + 1] = 0;     // the unused half of the last word
    }
    if (tid == 0) ws[g.off_holes + tile] = s_holes;
    if (x < g.w) {
#pragma unroll
        for (int r = 0; r < PROJ_TH / 4; ++r) {
            const int y = oy0 + wave + r * 4;
            if (y >= g.h) continue;
            float* o = out + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
            o[0] = vxv[r];
            o[s1.c] = vyv[r];
            count[(int64_t)b * sc.b + (int64_t)y * sc.h + x] = cv[r];
        }
    }
}

// K2: pass 3 (flowprojection_cuda_kernel.cu:175-231).  A cell read here is either a non-hole
// (never written by this pass) or is multiplied by 0.
//
// The reference walks cell by cell from every hole until it meets a non-zero count; along an
// uncovered border strip that is a dependent chain of up to H (or W) loads per hole.  On the
// normal path K1 has left row-packed and column-packed bitmaps of "count != 0", so a walk is a
// few word loads and a count-leading/trailing-zeros; only the cell found is then read.  The cell
// found -- hence the result -- is the same.  Tiles without holes (K1 counted them) leave at once.
struct ProjScan { int pos; float cnt; };

// cell-by-cell walk of the reference (fallback path: no bitmaps)
__device__ __forceinline__ ProjScan proj_walk_plain(const float* __restrict__ cn, int64_t origin, int64_t stride,
                                                    int p0, int len, int dir) {
    ProjScan r{p0, 0.0f};
    while (r.cnt == 0.0f && r.pos + dir >= 0 && r.pos + dir <= len - 1) { r.pos += dir; r.cnt = cn[origin + (int64_t)r.pos * stride]; }
    return r;
}

__global__ __launch_bounds__(PROJ_THREADS) void proj_finish(
    float* __restrict__ count, float* out, ProjGeom g, vfi_strides s1, vfi_strides sc,
    int* __restrict__ ws, const int* __restrict__ bits, const float* __restrict__ planes, int fillhole) {
    const int tile = blockIdx.x;
    const bool fallback = ws[PROJ_WS_DIRTY] != 0;
    if (tile == 0 && threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(ws + PROJ_WS_COST) = 0ull;   // K0 of the next call sums from 0
    if (!fallback && (!fillhole || ws[g.off_holes + tile] == 0)) return;
    const int per_img = g.tiles_x * g.tiles_y;
    const int b = tile / per_img;
    const int trem = tile - b * per_img;
    const int tyi = trem / g.tiles_x, txi = trem - tyi * g.tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = txi * PROJ_TW + lane;
    if (x >= g.w) return;
    float* o0 = out + (int64_t)b * s1.b;
    float* o1 = o0 + s1.c;
#pragma unroll 1
    for (int r = 0; r < PROJ_TH / 4; ++r) {
        const int y = tyi * PROJ_TH + wave + r * 4;
        if (y >= g.h) continue;
        const int64_t row = (int64_t)y * s1.h;
        if (fallback) {
            // K1 left sums in the scratch planes: normalise (pass 2) and fill holes (pass 3) from them
            const int64_t npx = (int64_t)(g.ntiles / per_img) * g.h * g.w;
            const float* p0 = planes + (int64_t)b * g.h * g.w;
            const float* p1 = p0 + npx;
            const float* pc = p1 + npx;
            const int64_t me = (int64_t)y * g.w + x;
            const float c = pc[me];
            count[(int64_t)b * sc.b + (int64_t)y * sc.h + x] = c;
            if (c > 0.0f) {
                o0[row + x] = p0[me] / c;
                o1[row + x] = p1[me] / c;
                continue;
            }
            float v0 = 0.0f, v1 = 0.0f;
            if (fillhole) {
                const ProjScan l = proj_walk_plain(pc, (int64_t)y * g.w, 1, x, g.w, -1), rr = proj_walk_plain(pc, (int64_t)y * g.w, 1, x, g.w, +1);
                const ProjScan u = proj_walk_plain(pc, x, g.w, y, g.h, -1), d = proj_walk_plain(pc, x, g.w, y, g.h, +1);
                if (l.cnt + rr.cnt + u.cnt + d.cnt > 0.0f) {
                    const float lt = (l.cnt > 0.0f) ? 1.0f : 0.0f, rt = (rr.cnt > 0.0f) ? 1.0f : 0.0f;
                    const float ut = (u.cnt > 0.0f) ? 1.0f : 0.0f, dt = (d.cnt > 0.0f) ? 1.0f : 0.0f;
                    const float den = lt + rt + ut + dt;
                    const int64_t il = (int64_t)y * g.w + l.pos, ir = (int64_t)y * g.w + rr.pos;
                    const int64_t iu = (int64_t)u.pos * g.w + x, id = (int64_t)d.pos * g.w + x;
                    // a neighbour found by a walk has count > 0; the others carry weight 0 (their cell is a hole: value 0)
                    const float a0 = l.cnt > 0.0f ? p0[il] / l.cnt : 0.0f, b0 = rr.cnt > 0.0f ? p0[ir] / rr.cnt : 0.0f;
                    const float c0 = u.cnt > 0.0f ? p0[iu] / u.cnt : 0.0f, d0 = d.cnt > 0.0f ? p0[id] / d.cnt : 0.0f;
                    const float a1 = l.cnt > 0.0f ? p1[il] / l.cnt : 0.0f, b1 = rr.cnt > 0.0f ? p1[ir] / rr.cnt : 0.0f;
                    const float c1 = u.cnt > 0.0f ? p1[iu] / u.cnt : 0.0f, d1 = d.cnt > 0.0f ? p1[id] / d.cnt : 0.0f;
                    v0 = (lt * a0 + rt * b0 + ut * c0 + dt * d0) / den;
                    v1 = (lt * a1 + rt * b1 + ut * c1 + dt * d1) / den;
                }
            }
            o0[row + x] = v0;
            o1[row + x] = v1;
            continue;
        }
        const float* cn = count + (int64_t)b * sc.b;
        if (!(cn[(int64_t)y * sc.h + x] <= 0.0f)) continue;
        // K1 ran its normal path and left the bitmaps
        const int* rl = bits + g.rowmap + (b * g.h + y) * g.rmw;
        const int* cl = bits + g.colmap + (b * g.w + x) * g.cmw;
        const int xl = proj_bit_walk(rl, x, g.w, -1), xr = proj_bit_walk(rl, x, g.w, +1);
        const int yu = proj_bit_walk(cl, y, g.h, -1), yd = proj_bit_walk(cl, y, g.h, +1);
        // a walk that found nothing contributes weight 0; its position only has to be valid
        ProjScan l, rr, u, d;
        l.pos = xl < 0 ? x : xl; rr.pos = xr < 0 ? x : xr; u.pos = yu < 0 ? y : yu; d.pos = yd < 0 ? y : yd;
        l.cnt = xl < 0 ? 0.0f : cn[(int64_t)y * sc.h + xl];
        rr.cnt = xr < 0 ? 0.0f : cn[(int64_t)y * sc.h + xr];
        u.cnt = yu < 0 ? 0.0f : cn[(int64_t)yu * sc.h + x];
        d.cnt = yd < 0 ? 0.0f : cn[(int64_t)yd * sc.h + x];
        if (l.cnt + rr.cnt + u.cnt + d.cnt <= 0.0f) continue;
        const float lt = (l.cnt > 0.0f) ? 1.0f : 0.0f;
        const float rt = (rr.cnt > 0.0f) ? 1.0f : 0.0f;
        const float ut = (u.cnt > 0.0f) ? 1.0f : 0.0f;
        const float dt = (d.cnt > 0.0f) ? 1.0f : 0.0f;
        const float den = lt + rt + ut + dt;
        o0[row + x] = (lt * o0[row + l.pos] + rt * o0[row + rr.pos] + ut * o0[(int64_t)u.pos * s1.h + x] +
                       dt * o0[(int64_t)d.pos * s1.h + x]) / den;
        o1[row + x] = (lt * o1[row + l.pos] + rt * o1[row + rr.pos] + ut * o1[(int64_t)u.pos * s1.h + x] +
                       dt * o1[(int64_t)d.pos * s1.h + x]) / den;
    }
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

// Geometry and workspace sizes of a call; false when the frame is beyond what the tables can index.
struct ProjSizes { size_t words, bit_words, plane_floats; };
static bool proj_geometry(int batch, int h, int w, ProjGeom* g, ProjSizes* z) {
    if (batch <= 0 || h <= 0 || w <= 0 || h > 32767 || w > 32767) return false;    // displacements are stored as int16
    g->h = h; g->w = w;
    g->tiles_x = (w + PROJ_TW - 1) / PROJ_TW;
    g->tiles_y = (h + PROJ_TH - 1) / PROJ_TH;
    const int64_t nt = (int64_t)g->tiles_x * g->tiles_y * batch;
    g->sup_x = (w + PROJ_SUP_W - 1) / PROJ_SUP_W;
    g->sup_y = (h + PROJ_SUP_H - 1) / PROJ_SUP_H;
    g->nsup = g->sup_x * g->sup_y;
    const int64_t ns = (int64_t)g->nsup * batch;
    if (nt > (1 << 24) || ns > (1 << 22)) return false;
    g->ntiles = (int)nt;
    g->rmw = (w + 31) / 32;
    g->cmw = (g->tiles_y * PROJ_TH + 31) / 32;              // whole tiles: K1 stores 16-bit halves
    z->bit_words = (size_t)batch * ((size_t)h * g->rmw + (size_t)w * g->cmw);
    if (z->bit_words > (size_t)INT_MAX) return false;
    g->rowmap = 0;
    g->colmap = batch * h * g->rmw;
    g->off_sup = PROJ_WS_HDR;
    g->off_blk = g->off_sup + 4 * (int)ns;
    g->off_holes = g->off_blk + 4 * PROJ_SUP_BLOCKS * (int)ns;
    z->words = (size_t)g->off_holes + (size_t)nt;
    z->plane_floats = (size_t)3 * batch * h * w;
    return true;
}

struct ProjBuffers { int* words; int* bits; float* planes; };
static bool proj_buffers(hipStream_t st, const ProjSizes& z, ProjBuffers* p) {
    // the header and the scratch planes carry state between calls and start at zero; the tables and the
    // bitmaps are rewritten by every call
    p->words = static_cast<int*>(ws_get(st, WS_PROJ_WORDS, z.words * sizeof(int), true, nullptr));
    p->bits = static_cast<int*>(ws_get(st, WS_PROJ_BITS, z.bit_words * sizeof(int), false, nullptr));
    p->planes = static_cast<float*>(ws_get(st, WS_PROJ_PLANES, z.plane_floats * sizeof(float), true, nullptr));
    return p->words && p->bits && p->planes;
}

// s1 = strides of `out` (the reference binding shares them with the input flow)
template <bool DEPTH, bool UP>
static int project_forward(const ProjFlow& flow, const float* in2, float* count, float* out, int batch, int h, int w,
                           int fillhole, vfi_strides s1, vfi_strides s2, vfi_strides sc, hipStream_t st) {
    ProjGeom g;
    ProjSizes z;
    if (!proj_geometry(batch, h, w, &g, &z)) return VFI_ERR_SHAPE;
    ProjBuffers p;
    if (!proj_buffers(st, z, &p)) return VFI_ERR_LAUNCH;
    const int nsup_all = g.nsup * batch;
    const unsigned long long cost_limit = (unsigned long long)PROJ_COST_LIMIT * PROJ_SUP_BLOCKS * (unsigned long long)nsup_all;
    hipLaunchKernelGGL((proj_scan<DEPTH, UP>), dim3(nsup_all), dim3(PROJ_SCAN_THREADS), 0, st, flow, in2, g, s2, p.words,
                       p.planes, (int64_t)z.plane_floats);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    hipLaunchKernelGGL((proj_pull<DEPTH, UP>), dim3(g.ntiles), dim3(PROJ_THREADS), 0, st, flow, in2, count, out, g, s1,
                       s2, sc, p.words, p.bits, p.planes, cost_limit);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    // (also runs with fillhole == 0: it resets the call's state, and the fallback path normalises there)
    hipLaunchKernelGGL(proj_finish, dim3(g.ntiles), dim3(PROJ_THREADS), 0, st, count, out, g, s1, sc, p.words, p.bits,
                       p.planes, fillhole);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    return VFI_OK;
}

// standalone x4 upsample of (m0 * in) * m1 -- forward_flownets as one launch
__global__ __launch_bounds__(VFI_TX * VFI_TY) void flow_upsample4(
    const float* __restrict__ in, float* __restrict__ out, int channels, int hq, int wq, float m0, float m1,
    vfi_strides sq, vfi_strides so) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= 4 * wq || y >= 4 * hq) return;
    const int b = blockIdx.z;
    const UpTap ty = up4_tap(y, hq), tx = up4_tap(x, wq);
    for (int c = 0; c < channels; ++c)
        out[(int64_t)b * so.b + (int64_t)c * so.c + (int64_t)y * so.h + x] =
            up4_sample(in + (int64_t)b * sq.b + (int64_t)c * sq.c, sq.h, ty, tx, m0, m1);
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_projection_reserve(int batch, int h, int w, vfi_stream_t stream) {
    ProjGeom g;
    ProjSizes z;
    if (!proj_geometry(batch, h, w, &g, &z)) return VFI_ERR_SHAPE;
    ProjBuffers p;
    return proj_buffers((hipStream_t)stream, z, &p) ? VFI_OK : VFI_ERR_LAUNCH;
}

extern "C" int vfi_flowprojection_forward(const float* input1, float* count, float* output, int batch, int h, int w,
                                           int fillhole, vfi_strides s1, vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !count || !output) return VFI_ERR_SHAPE;
    const ProjFlow flow{input1, s1, 0, 0, 1.0f, 1.0f};
    return project_forward<false, false>(flow, nullptr, count, output, batch, h, w, fillhole, s1, s1, sc, (hipStream_t)stream);
}

extern "C" int vfi_depthflowprojection_forward(const float* input1, const float* input2, float* count, float* output,
                                                int batch, int h, int w, int fillhole, vfi_strides s1, vfi_strides s2,
                                                vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !count || !output) return VFI_ERR_SHAPE;
    const ProjFlow flow{input1, s1, 0, 0, 1.0f, 1.0f};
    return project_forward<true, false>(flow, input2, count, output, batch, h, w, fillhole, s1, s2, sc, (hipStream_t)stream);
}

// ---- fused glue: the network's quarter-resolution flow goes straight into the splat
extern "C" int vfi_flow_upsample4(const float* input, float* output, int batch, int channels, int hq, int wq,
                                   float mul0, float mul1, vfi_strides sq, vfi_strides so, vfi_stream_t stream) {
    if (batch <= 0 || channels <= 0 || hq <= 0 || wq <= 0 || hq > INT_MAX / 4 || wq > INT_MAX / 4 || !input || !output)
        return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(flow_upsample4, pixel_grid(4 * wq, 4 * hq, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input, output, channels, hq, wq, mul0, mul1, sq, so);
    return launch_status();
}

extern "C" int vfi_flowprojection_forward_up4(const float* flow_q, float* count, float* output, int batch, int hq, int wq,
                                               float mul0, float mul1, int fillhole, vfi_strides sq, vfi_strides sc,
                                               vfi_strides so, vfi_stream_t stream) {
    if (batch <= 0 || hq <= 0 || wq <= 0 || hq > INT_MAX / 4 || wq > INT_MAX / 4 || !flow_q || !count || !output)
        return VFI_ERR_SHAPE;
    const ProjFlow flow{flow_q, sq, hq, wq, mul0, mul1};
    return project_forward<false, true>(flow, nullptr, count, output, batch, 4 * hq, 4 * wq, fillhole, so, so, sc,
                                        (hipStream_t)stream);
}

extern "C" int vfi_depthflowprojection_forward_up4(const float* flow_q, const float* input2, float* count, float* output,
                                                    int batch, int hq, int wq, float mul0, float mul1, int fillhole,
                                                    vfi_strides sq, vfi_strides s2, vfi_strides sc, vfi_strides so,
                                                    vfi_stream_t stream) {
    if (batch <= 0 || hq <= 0 || wq <= 0 || hq > INT_MAX / 4 || wq > INT_MAX / 4 || !flow_q || !input2 || !count || !output)
        return VFI_ERR_SHAPE;
    const ProjFlow flow{flow_q, sq, hq, wq, mul0, mul1};
    return project_forward<true, true>(flow, input2, count, output, batch, 4 * hq, 4 * wq, fillhole, so, s2, sc,
                                       (hipStream_t)stream);
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
