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
// OUTPUT tiles and, as a source, into 16x16 BLOCKS.
//   K0  proj_scan4   reads the flow once: per block the range of its integer displacements and its largest
//                    addends; for every output tile the block can reach, the part of the block that can reach
//                    it is merged into that tile's SOURCE RECTANGLE (a 32-byte record, fields merged by max:
//                    first in a table of the workgroup in LDS, then by a few atomicMax instructions).
//   K1  proj_pull_lean  one workgroup per output tile: walks its source rectangle (for a smooth field: the tile
//                    shifted by the flow and a few pixels larger, ~1.4 source pixels per output pixel),
//                    accumulates in LDS and writes count and the normalised flow once, coalesced, plus two
//                    bitmaps of "count != 0" and whether the tile has holes.
//   K2  proj_finish  hole filling for the tiles that have holes (bitmap searches in registers instead of the
//                    reference's cell-by-cell walks); resets the per-call state.
// (proj_scan / proj_pull: the same two steps with 4-byte lanes, for flows whose rows are not 16-byte aligned.)
// K1 writes every cell of count and output, so callers need not zero-fill them (the reference's
// callers must: its splat accumulates into them).  Nothing of a call's state crosses to the host and
// no kernel argument changes from call to call: a captured graph is replayable.
//
// Fallback: a block whose pixels scatter over more than PROJ_BLOCK_CAP output tiles (random flow of
// +-W/2) raises a flag in K0; K1 then splats its own tile with global atomics exactly like the
// reference -- into three scratch planes of the workspace that are zero between calls -- and K2
// normalises them into count / output and fills holes from them.  (The scratch planes are cleaned by
// the next call's K0, whose workgroups see a "dirty" word.)
//
// Accumulation in K1 is fixed point in LDS integer atomics: on gfx950 an LDS float atomic add costs
// ~170 cycles per wave instruction (tools/probes/lds_atomic_probe.hip), an integer one a few.  Every
// addend is scaled by a power of two chosen per output tile from the largest |addend| that can reach it
// (K0) so that it is below 2^25, rounded to an integer, and the two flow components are added as ONE
// 64-bit integer (x << 32) + y: with at most 32 addends per cell neither half leaves its 32 bits, and
// the halves are separated exactly afterwards.  The number of addends per cell is accumulated beside it
// (it IS the count plane of FlowProjection); a tile with a busier cell (flows converging 8-fold) is
// accumulated again with a correspondingly coarser scale.  Sums are exact integers, so the result does
// not depend on the summation order -- reproducible bit for bit from run to run (the reference's fp32
// atomic sum carries one rounding per addend, in arrival order) -- and agrees with any fp32 summation
// order to rounding; addends that are multiples of 2^-k (k < ~16) sum exactly in both.  count of
// FlowProjection is exact.
#include "vfi_common.h"
#include "bitwalk.h"
#include "workspace.h"

#ifndef PROJ_DEV_SKIP
#define PROJ_DEV_SKIP 0
#endif

#include <limits.h>

#include <type_traits>

namespace vfi {

#define PROJ_TW 64                  // output tile
#define PROJ_TH 16
#define PROJ_THREADS 256
#define PROJ_BLK 16                 // source block edge
#define PROJ_ADD_BITS 25            // |scaled addend| < 2^25
#define PROJ_ADD_CELL 32            // addends per cell that fit beside it in 32 bits
#define PROJ_CLS_BITS 6             // binary orders of magnitude of weight per accumulation pass (DepthFlowProjection)
#define PROJ_BLOCK_CAP 64           // output tiles one block may reach before the call takes the fallback
#define PROJ_STAB 16                // K0's table of merged record updates: 16 x 16 tiles around the workgroup's four,
#define PROJ_STAB_X0 6              // from 6 tile columns (384 pixels) left of them
#define PROJ_STAB_Y0 8              // and 8 tile rows (128 pixels) above

// workspace "words" (32-bit).  Header: [0] a block of this call reaches too many tiles: fallback (set by K0,
// read by K1, reset by K2); [2] the scratch planes of the fallback hold sums (written by K1, read by K2 and by
// the next call's K0).
// Then one 8-word record per output tile: its source rectangle as 32767 - x0, 32767 - y0, x1 + 1, y1 + 1, the bits of the
// largest |fx|, of the largest |count addend| and (inverted) of the smallest weight that reach it, and the bits of the largest
// |fy| -- all merged with atomicMax by K0, so 0 = nothing; K1 reads its record and zeroes it again.  Then one word per tile from K1 to K2: has it holes.
// workspace "bits": two bitmaps of "count != 0", one packed along rows (rowmap[b][y][x/32]) and one packed
// along columns (colmap[b][x][y/32], lines padded to whole 16-byte groups), written by K1 for the hole filler.
#define PROJ_WS_HDR 16
#define PROJ_WS_FLAG 0
#define PROJ_WS_DIRTY 2
#define PROJ_TILE_WORDS 8
#define PROJ_INV_BITS 0x7f7fffff       // a weight's bits are stored as this minus them where the SMALLEST is wanted

// header words [2], [3]: how many floats of the fallback's scratch planes hold sums (64 bits; 0 = clean).  The count, not a
// flag: the next call cleans what THIS call dirtied, whatever its own frame size is.
__device__ __forceinline__ int64_t proj_dirty_floats(const int* ws) {
    return (int64_t)(unsigned)ws[PROJ_WS_DIRTY] | ((int64_t)(unsigned)ws[PROJ_WS_DIRTY + 1] << 32);
}

// rmw / cmw: 32-bit words per image row / column of the two bitmaps; rowmap / colmap: their word offsets
// inside the bit buffer; off_tile / off_list: word offsets of the tile records and of the hole list
struct ProjGeom {
    int h, w, tiles_x, tiles_y, ntiles, rmw, cmw, rowmap, colmap, off_tile, off_list;
};

// The flow (and depth) planes a projection reads: the full-resolution flow tensor of the reference's FlowProjection.
// (The network's quarter-resolution flow goes through flow_upsample4 first: vfi_*_forward_up4 below.)
// Row strides are 32-bit here (the host checks that every in-plane offset fits): the kernels address a
// plane as uniform base + 32-bit offset.
// A call projects a LIST of flows (the networks' FlowProject(inputs, depth): networks/DAIN.py:533-539,
// networks/DAIN_slowmotion.py:301-307 -- one launch triple for the whole list instead of one per flow): item i < n has its own
// flow / depth / count / output tensors, all items share shape and strides.  The kernels see n * per "images"; image b is
// image b % per of item b / per.  (The pointer tables are kernel arguments, by value: a captured graph replays them.)
#define PROJ_NMAX 8
struct ProjSrc {
    const float* flow[PROJ_NMAX];       // [per,2,h,w] each
    const float* depth[PROJ_NMAX];      // [per,1,h,w] each (DEPTH only; items may share one)
    int64_t fb, fc, db;     // flow batch / channel stride, depth batch stride
    int fh, dh;             // row strides
    int per;                // images per item
};
struct ProjDst {
    float* count[PROJ_NMAX];            // [per,1,h,w] each
    float* out[PROJ_NMAX];              // [per,2,h,w] each
};
struct ProjImage { int item, bi; };
__device__ __forceinline__ ProjImage proj_image(int b, int per) {
    if (per == 1) return ProjImage{b, 0};
    const int it = b / per;
    return ProjImage{it, b - it * per};
}

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

// A source pixel's raw values: pix_load only issues loads, so that a caller can have the loads of many pixels in flight
// before it works on the first.
// Addressing is buffer-style: a wave-uniform descriptor per plane + a 32-bit byte offset split into a
// per-lane part and a wave-uniform part (the row a wave works on advances on the scalar unit, at no
// vector instruction per pixel); offsets past the plane read as 0.
struct ProjPix { float fx, fy, d; };
struct ProjPlanes { __amdgpu_buffer_rsrc_t f0, f1, d; };

template <bool DEPTH>
__device__ __forceinline__ ProjPlanes proj_planes(const ProjSrc& s, int b, int h, int w) {
    const ProjImage im = proj_image(b, s.per);
    const float* f0 = s.flow[im.item] + (int64_t)im.bi * s.fb;
    ProjPlanes p;
    p.f0 = __builtin_amdgcn_make_buffer_rsrc((void*)f0, 0, ((h - 1) * s.fh + w) * 4, 0x00020000);
    p.f1 = __builtin_amdgcn_make_buffer_rsrc((void*)(f0 + s.fc), 0, ((h - 1) * s.fh + w) * 4, 0x00020000);
    p.d = p.f0;
    if constexpr (DEPTH)
        p.d = __builtin_amdgcn_make_buffer_rsrc((void*)(s.depth[im.item] + (int64_t)im.bi * s.db), 0, ((h - 1) * s.dh + w) * 4, 0x00020000);
    return p;
}
__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// pixel (x, yl + yu): x and yl may differ from lane to lane, yu is wave-uniform.  Inside the frame.
template <bool DEPTH>
__device__ __forceinline__ ProjPix pix_load(const ProjSrc& s, const ProjPlanes& p, int x, int yl, int yu) {
    ProjPix r;
    const int vo = (yl * s.fh + x) * 4, so = yu * s.fh * 4;
    r.fx = buf_f32(p.f0, vo, so);
    r.fy = buf_f32(p.f1, vo, so);
    r.d = 1.0f;
    if constexpr (DEPTH) r.d = buf_f32(p.d, (yl * s.dh + x) * 4, yu * s.dh * 4);
    return r;
}
// The reference's test 0 <= x2 <= w - 1 (flowprojection_cuda_kernel.cu:69) as ONE unsigned comparison of
// the float's bits with those of (float)(w - 1): non-negative floats order like their bit patterns, negative
// ones (sign bit) and NaNs compare above every finite positive one.  (x2 = x + fx with x >= +0 is never -0.)
__device__ __forceinline__ bool pix_target(float fx, float fy, int x, int y, unsigned wbits, unsigned hbits, int& L, int& T) {
    const float x2 = (float)x + fx, y2 = (float)y + fy;
    L = (int)x2;
    T = (int)y2;
    return (unsigned)__float_as_int(x2) <= wbits && (unsigned)__float_as_int(y2) <= hbits;
}

// Workgroup -> work item.  Workgroups are dealt round-robin over the 8 XCDs (observed on gfx950: block b runs on XCD
// b % 8 in every launch; speed only -- any dealing gives the same results), each XCD with an L2 of its own.  Item =
// the b/8-th of XCD b%8's contiguous eighth of the raster order: an XCD owns one horizontal band of the frame in
// K0, K1 and K2 alike, so the flow K0 has just read is in the L2 that K1's workgroups read their source
// rectangles through, and the counts, flows and bitmaps K1 writes are in the L2 K2 searches them through.  A
// bijection of [0, n) for every n.
#ifndef PROJ_BANDS
#define PROJ_BANDS 1
#endif
__device__ __forceinline__ int band_item(int bid, int n) {
#if PROJ_BANDS
    const int x = bid & 7, k = bid >> 3, base = n >> 3, rem = n & 7;
    return x * base + min(x, rem) + k;
#else
    return bid;
#endif
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

// The tail of K0.  Lanes 15, 31, 47, 63 of the wave speak for four 16x16 blocks that sit side by side (x = the
// block's last column, y0 = its first row) and mostly feed the same two or three output tiles: per tile of the
// union of their ranges the four parts are merged in registers and the record is updated by ONE atomic
// instruction (lane k = field k) -- the atomics are the expensive part of this kernel (one wave instruction
// per ~50 ns per CU at the memory side, whatever its lane count).
__device__ __forceinline__ void scan_scatter(const ProjGeom& g, int* __restrict__ ws, int b, int lane, int x, int y0,
                                             int dlmin, int dlmax, int dtmin, int dtmax, int vbits, int cbits, int mbits, int fbits) {
    const bool any = dlmin != INT_MAX;
    const int bx0 = x - 15, bx1 = min(x, g.w - 1), by1 = min(y0 + PROJ_BLK - 1, g.h - 1);
    // top-left targets of the block lie in [X0, X1] x [Y0, Y1]; a target (L, T) feeds columns L, L + 1, rows T, T + 1
    const int X0 = max(bx0 + dlmin, 0), X1 = min(bx1 + dlmax, g.w - 1);
    const int Y0 = max(y0 + dtmin, 0), Y1 = min(by1 + dtmax, g.h - 1);
    int ta = INT_MAX, tb = INT_MIN, tc = INT_MAX, td = INT_MIN;
    bool wild = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = 16 * q + 15;
        if (!__builtin_amdgcn_readlane((int)any, sl)) continue;
        const int a0 = __builtin_amdgcn_readlane(X0, sl) / PROJ_TW, a1 = min(__builtin_amdgcn_readlane(X1, sl) + 1, g.w - 1) / PROJ_TW;
        const int c0 = __builtin_amdgcn_readlane(Y0, sl) / PROJ_TH, c1 = min(__builtin_amdgcn_readlane(Y1, sl) + 1, g.h - 1) / PROJ_TH;
        wild = wild || (a1 - a0 + 1) * (c1 - c0 + 1) > PROJ_BLOCK_CAP;
        ta = min(ta, a0); tb = max(tb, a1); tc = min(tc, c0); td = max(td, c1);
    }
    if (wild) { if (lane == 0) ws[PROJ_WS_FLAG] = 1; return; }
    for (int ty = tc; ty <= td; ++ty)
        for (int tx = ta; tx <= tb; ++tx) {
            const int ox0 = tx * PROJ_TW, oy0 = ty * PROJ_TH;
            const int tx1 = min(ox0 + PROJ_TW - 1, g.w - 1), ty1 = min(oy0 + PROJ_TH - 1, g.h - 1);
            // a pixel at x has L in [x + dlmin, x + dlmax]; the tile takes L in [ox0 - 1, tx1]
            const int sx0 = max(bx0, ox0 - 1 - dlmax), sx1 = min(bx1, tx1 - dlmin);
            const int sy0 = max(y0, oy0 - 1 - dtmax), sy1 = min(by1, ty1 - dtmin);
            const bool hit = any && sx0 <= sx1 && sy0 <= sy1;
            // fields as stored (0 = nothing), merged over the four blocks
            int f0m = 0, f1m = 0, f2m = 0, f3m = 0, f4m = 0, f5m = 0, f6m = 0, f7m = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int sl = 16 * q + 15;
                if (!__builtin_amdgcn_readlane((int)hit, sl)) continue;
                f0m = max(f0m, 32767 - __builtin_amdgcn_readlane(sx0, sl)); f1m = max(f1m, 32767 - __builtin_amdgcn_readlane(sy0, sl));
                f2m = max(f2m, __builtin_amdgcn_readlane(sx1, sl) + 1); f3m = max(f3m, __builtin_amdgcn_readlane(sy1, sl) + 1);
                f4m = max(f4m, __builtin_amdgcn_readlane(vbits, sl)); f5m = max(f5m, __builtin_amdgcn_readlane(cbits, sl));
                f6m = max(f6m, __builtin_amdgcn_readlane(mbits, sl));
                f7m = max(f7m, __builtin_amdgcn_readlane(fbits, sl));
            }
            if (f2m == 0) continue;
            int* e = ws + g.off_tile + (int64_t)((b * g.tiles_y + ty) * g.tiles_x + tx) * PROJ_TILE_WORDS;
            const int v = lane == 0 ? f0m : lane == 1 ? f1m : lane == 2 ? f2m : lane == 3 ? f3m : lane == 4 ? f4m : lane == 5 ? f5m : lane == 6 ? f6m : f7m;
            if (lane < 8) atomicMax(&e[lane], v);
        }
}

// K0: one wave per 64 x 16 pixels of the SOURCE frame (lane = x, 16 rows; all loads of the wave in flight at
// once).  Per 16x16 block: the range of (L - x) and of (T - y) over its valid pixels; from it the output tiles the
// block can reach, and for each of them the part of the block that can -- merged into that tile's source
// rectangle with atomicMax (fields stored so that 0 means "nothing").
template <bool DEPTH>
__global__ __launch_bounds__(64) void proj_scan(ProjSrc src, ProjGeom g, int* __restrict__ ws, float* __restrict__ planes,
                                                int64_t plane_floats) {
    const int64_t dirty = proj_dirty_floats(ws);
    if (dirty != 0) {
        // the previous call on this workspace took the fallback: the floats IT dirtied (its frame may have been larger
        // than this one) are cleaned here, a slice per workgroup (this call's K1 rewrites the words)
        const int64_t chunk = (dirty + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(dirty, lo + chunk);
        for (int64_t i = lo + threadIdx.x; i < hi; i += 64) planes[i] = 0.0f;
    }
    const int tile = band_item(blockIdx.x, gridDim.x), lane = threadIdx.x;
    const int per_img = g.tiles_x * g.tiles_y;
    const int b = tile / per_img;
    const int trem = tile - b * per_img;
    const int tyi = trem / g.tiles_x, txi = trem - tyi * g.tiles_x;
    const int x = txi * PROJ_TW + lane, y0 = tyi * PROJ_TH;
    const int xc = min(x, g.w - 1);
    const ProjPlanes pl = proj_planes<DEPTH>(src, b, g.h, g.w);
    const unsigned wbits = (unsigned)__float_as_int((float)(g.w - 1)), hbits = (unsigned)__float_as_int((float)(g.h - 1));
    int dlmin = INT_MAX, dlmax = INT_MIN, dtmin = INT_MAX, dtmax = INT_MIN, vbits = 0, cbits = 0, mbits = 0, fbits = 0;
    constexpr int GROUP = PROJ_TH;                          // rows whose loads are issued together
#pragma unroll 1
    for (int r0 = 0; r0 < PROJ_TH; r0 += GROUP) {
        ProjPix raw[GROUP];
#pragma unroll
        for (int k = 0; k < GROUP; ++k) raw[k] = pix_load<DEPTH>(src, pl, xc, 0, min(y0 + r0 + k, g.h - 1));
#pragma unroll
        for (int k = 0; k < GROUP; ++k) {
            const int y = y0 + r0 + k;
            const float fx = raw[k].fx, fy = raw[k].fy;
            int L, T;
            const bool valid = pix_target(fx, fy, x, y, wbits, hbits, L, T) && x < g.w && y < g.h;
            const int dl = L - x, dt = T - y;
            dlmin = min(dlmin, valid ? dl : INT_MAX); dlmax = max(dlmax, valid ? dl : INT_MIN);
            dtmin = min(dtmin, valid ? dt : INT_MAX); dtmax = max(dtmax, valid ? dt : INT_MIN);
            // the largest |fx| and |fy| (each component gets its own fixed-point scale) and, with depth, the largest and the
            // smallest |weight| (the latter stored inverted, so that "largest" merges it); non-negative floats order like
            // their bit patterns
            vbits = max(vbits, valid ? __float_as_int(fabsf(fx)) : 0);
            fbits = max(fbits, valid ? __float_as_int(fabsf(fy)) : 0);
            if constexpr (DEPTH) {
                const int db = __float_as_int(fabsf(raw[k].d));
                cbits = max(cbits, valid ? db : 0);
                mbits = max(mbits, valid && db != 0 ? PROJ_INV_BITS - db : 0);
            }
        }
    }
    dlmin = row16_min(dlmin); dlmax = row16_max(dlmax);
    dtmin = row16_min(dtmin); dtmax = row16_max(dtmax);
    vbits = row16_max(vbits); cbits = row16_max(cbits); mbits = row16_max(mbits); fbits = row16_max(fbits);
    scan_scatter(g, ws, b, lane, x, y0, dlmin, dlmax, dtmin, dtmax, vbits, cbits, mbits, fbits);
}

// K0 for a full-resolution flow: the same, with 16-byte lanes (a walk with 4-byte lanes reaches ~4 TB/s on this
// chip, one with 16-byte lanes ~5.3).  A workgroup of four waves covers 256 x 16 pixels -- four tiles side by
// side; a lane owns four consecutive pixels, a wave four rows; the 16x16 blocks' ranges are combined across
// the waves in LDS, then wave w scatters the four blocks of tile column w exactly as above.
typedef int proj_v4i __attribute__((ext_vector_type(4)));
typedef float proj_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ proj_v4f buf_f32x4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(proj_v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
#define PROJ_QUAD_STEP(OP, SEL) v = OP(v, __builtin_amdgcn_update_dpp(v, v, SEL, 0xf, 0xf, false))
__device__ __forceinline__ int quad_min(int v) { PROJ_QUAD_STEP(min, 0xb1); PROJ_QUAD_STEP(min, 0x4e); return v; }   // quad_perm [1,0,3,2], [2,3,0,1]
__device__ __forceinline__ int quad_max(int v) { PROJ_QUAD_STEP(max, 0xb1); PROJ_QUAD_STEP(max, 0x4e); return v; }
#undef PROJ_QUAD_STEP

template <bool DEPTH>
__global__ __launch_bounds__(256) void proj_scan4(ProjSrc src, ProjGeom g, int groups_x, int* __restrict__ ws,
                                                  float* __restrict__ planes, int64_t plane_floats) {
    __shared__ int sblk[16][8];                             // per block: dlmin, dlmax, dtmin, dtmax, vbits, cbits, mbits, fbits
    __shared__ __attribute__((aligned(16))) int stab[PROJ_STAB * PROJ_STAB][8];   // the workgroup's merged updates: [tile slot][record field]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 128) sblk[tid >> 3][tid & 7] = ((tid & 7) == 0 || (tid & 7) == 2) ? INT_MAX : ((tid & 7) == 1 || (tid & 7) == 3) ? INT_MIN : 0;
    reinterpret_cast<uint4*>(&stab[tid][0])[0] = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(&stab[tid][0])[1] = make_uint4(0u, 0u, 0u, 0u);
    const int per_img = groups_x * g.tiles_y;
    const int item = band_item(blockIdx.x, gridDim.x);
    const int b = item / per_img;
    const int rem = item - b * per_img;
    const int tyi = rem / groups_x, gxi = rem - tyi * groups_x;
    const int x0 = gxi * 4 * PROJ_TW + 4 * lane, y0 = tyi * PROJ_TH, yw = y0 + 4 * wave;
    const ProjPlanes pl = proj_planes<DEPTH>(src, b, g.h, g.w);
    const unsigned wbits = (unsigned)__float_as_int((float)(g.w - 1)), hbits = (unsigned)__float_as_int((float)(g.h - 1));
    proj_v4f qx[4], qy[4], qd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int so = min(yw + r, g.h - 1) * src.fh * 4;
        qx[r] = buf_f32x4(pl.f0, x0 * 4, so);
        qy[r] = buf_f32x4(pl.f1, x0 * 4, so);
        if constexpr (DEPTH) qd[r] = buf_f32x4(pl.d, x0 * 4, min(yw + r, g.h - 1) * src.dh * 4);
    }
    // (behind the flow loads, so that they are in flight while the header word arrives)
    const int64_t dirty = proj_dirty_floats(ws);
    if (dirty != 0) {                                       // (see proj_scan)
        const int64_t chunk = (dirty + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(dirty, lo + chunk);
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) planes[i] = 0.0f;
    }
    __syncthreads();
    int dlmin = INT_MAX, dlmax = INT_MIN, dtmin = INT_MAX, dtmax = INT_MIN, vbits = 0, cbits = 0, mbits = 0, fbits = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j, y = yw + r;
            const float fx = qx[r][j], fy = qy[r][j];
            int L, T;
            const bool valid = pix_target(fx, fy, x, y, wbits, hbits, L, T) && x < g.w && y < g.h;
            if (valid) {
                dlmin = min(dlmin, L - x); dlmax = max(dlmax, L - x);
                dtmin = min(dtmin, T - y); dtmax = max(dtmax, T - y);
                // the largest |fx| and |fy|, the largest and (inverted) the smallest |weight|; non-negative floats order
                // like their bit patterns
                vbits = max(vbits, __float_as_int(fabsf(fx)));
                fbits = max(fbits, __float_as_int(fabsf(fy)));
                if constexpr (DEPTH) {
                    const int db = __float_as_int(fabsf(qd[r][j]));
                    cbits = max(cbits, db);
                    if (db != 0) mbits = max(mbits, PROJ_INV_BITS - db);
                }
            }
        }
#if PROJ_DEV_SKIP == 6      // (development: K0 = its loads and per-pixel arithmetic only)
    if (dlmin != 12345) { if (vbits == 0x7fffffff) ws[4] = dlmax + dtmin + dtmax + cbits + mbits; return; }
#endif
    // a block is four lanes wide
    dlmin = quad_min(dlmin); dlmax = quad_max(dlmax); dtmin = quad_min(dtmin); dtmax = quad_max(dtmax);
    vbits = quad_max(vbits); cbits = quad_max(cbits); mbits = quad_max(mbits); fbits = quad_max(fbits);
    if ((lane & 3) == 0 && dlmin != INT_MAX) {
        int* e = sblk[lane >> 2];
        atomicMin(&e[0], dlmin); atomicMax(&e[1], dlmax); atomicMin(&e[2], dtmin); atomicMax(&e[3], dtmax);
        atomicMax(&e[4], vbits); atomicMax(&e[5], cbits); atomicMax(&e[6], mbits); atomicMax(&e[7], fbits);
    }
    __syncthreads();
    // The tail (round 3).  Thread (block k, slot j): block k of the workgroup's 16 and the j-th of the <= 2 x 8 output tiles it
    // can reach; the part of the block that can reach that tile goes, field by field, into a table of the workgroup's
    // merged updates in LDS (cheap integer atomics); after a barrier the table's non-empty (tile, field) pairs go to the
    // records with ONE atomic instruction per eight tiles -- ~6 vector-memory atomic instructions per workgroup where
    // the per-wave readlane loops of scan_scatter issued ~24 (and ~700 scalar instructions per wave: 5 of this
    // kernel's 11 us at 1080p).
    {
        const int k = tid >> 4, j = tid & 15;
        const int* e = sblk[k];
        const int bx0 = gxi * 4 * PROJ_TW + k * PROJ_BLK, bx1 = min(bx0 + PROJ_BLK - 1, g.w - 1), by1 = min(y0 + PROJ_BLK - 1, g.h - 1);
        const int dlmin_ = e[0], dlmax_ = e[1], dtmin_ = e[2], dtmax_ = e[3];
        const bool any = dlmin_ != INT_MAX && bx0 < g.w;
#if PROJ_DEV_SKIP == 5      // (development: K0 without its scatter)
        if (dlmin_ != 12345) return;
#endif
        // top-left targets of the block lie in [X0, X1] x [Y0, Y1]; a target (L, T) feeds columns L, L + 1, rows T, T + 1
        const int X0 = max(bx0 + dlmin_, 0), X1 = min(bx1 + dlmax_, g.w - 1);
        const int Y0 = max(y0 + dtmin_, 0), Y1 = min(by1 + dtmax_, g.h - 1);
        const int a0 = X0 / PROJ_TW, a1 = min(X1 + 1, g.w - 1) / PROJ_TW, c0 = Y0 / PROJ_TH, c1 = min(Y1 + 1, g.h - 1) / PROJ_TH;
        const int nx = a1 - a0 + 1, ny = c1 - c0 + 1;
        const bool wild = any && nx * ny > PROJ_BLOCK_CAP;
        if (wild && j == 0) ws[PROJ_WS_FLAG] = 1;               // the call takes the fallback; records no longer matter
        const bool wide = any && !wild && (nx > 2 || ny > 8);   // more tiles than slots: thread j == 0 walks them all
        const int torg_x = gxi * 4 - PROJ_STAB_X0, torg_y = tyi - PROJ_STAB_Y0;
        auto update = [&](int tx, int ty) {
            const int ox0 = tx * PROJ_TW, oy0 = ty * PROJ_TH;
            const int tx1 = min(ox0 + PROJ_TW - 1, g.w - 1), ty1 = min(oy0 + PROJ_TH - 1, g.h - 1);
            // a pixel at x has L in [x + dlmin, x + dlmax]; the tile takes L in [ox0 - 1, tx1]
            const int sx0 = max(bx0, ox0 - 1 - dlmax_), sx1 = min(bx1, tx1 - dlmin_);
            const int sy0 = max(y0, oy0 - 1 - dtmax_), sy1 = min(by1, ty1 - dtmin_);
            if (sx0 > sx1 || sy0 > sy1) return;
            const int f[8] = {32767 - sx0, 32767 - sy0, sx1 + 1, sy1 + 1, e[4], e[5], e[6], e[7]};
            const int rx = tx - torg_x, ry = ty - torg_y;
            if ((unsigned)rx < PROJ_STAB && (unsigned)ry < PROJ_STAB) {
#pragma unroll
                for (int q = 0; q < 8; ++q) atomicMax(&stab[ry * PROJ_STAB + rx][q], f[q]);
            } else {                                            // beyond the table (flows of hundreds of pixels): straight to the record
                int* rec = ws + g.off_tile + (int64_t)((b * g.tiles_y + ty) * g.tiles_x + tx) * PROJ_TILE_WORDS;
#pragma unroll
                for (int q = 0; q < 8; ++q) atomicMax(&rec[q], f[q]);
            }
        };
        if (any && !wild && !wide) {
            const int tx = a0 + (j & 1), ty = c0 + (j >> 1);
            if (tx <= a1 && ty <= c1) update(tx, ty);
        } else if (wide && j == 0) {
            for (int ty = c0; ty <= c1; ++ty)
                for (int tx = a0; tx <= a1; ++tx) update(tx, ty);
        }
        __syncthreads();
        // lane = (tile slot, field): eight slots per wave instruction
#pragma unroll
        for (int pass = 0; pass < PROJ_STAB * PROJ_STAB / 32; ++pass) {
            const int sl = 32 * pass + (tid >> 3), q = tid & 7;
            const int v = stab[sl][q];
            if (v != 0) {
                const int tx = torg_x + (sl % PROJ_STAB), ty = torg_y + (sl / PROJ_STAB);
                atomicMax(ws + g.off_tile + (int64_t)((b * g.tiles_y + ty) * g.tiles_x + tx) * PROJ_TILE_WORDS + q, v);
            }
        }
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

// K1: one workgroup per output tile.  Each source pixel of the tile's rectangle adds its addend ONCE, at its
// top-left target, into a (16 + 1) x (64 + 1) grid (one row above and one column left of the tile included);
// a cell of the tile is then the sum of the 2 x 2 grid cells whose splats cover it (integer sums: exact, so
// this equals adding every addend to its four targets).  At the last column / row of the frame R == L / B == T:
// the reference adds twice there (flowprojection_cuda_kernel.cu:72-73), so does the sum below.
//
// Workgroup size: every tile costs the same and all start within a microsecond, so what matters is that ALL
// tiles are resident at once (a 1080p frame has 2232 tiles; 8 workgroups of 256 threads per CU would hold
// 2048 and leave a second round).  128 threads per tile: 9 workgroups per CU within the LDS.
// Memory: the kernel is bound by the memory system (in-kernel stamps, tools/proj_stamps.py: the later a
// workgroup's requests are queued, the longer it lives), so the flow is read and count / output are written
// with 16-byte lanes: a lane owns four consecutive pixels.
#ifndef PROJ_PULL_THREADS
#define PROJ_PULL_THREADS 128
#endif
#ifndef PROJ_PULL_WAVES
#define PROJ_PULL_WAVES 5           // 9 workgroups x 2 waves per CU
#endif
#ifndef PROJ_PULL_CH
#define PROJ_PULL_CH 2              // loads in flight per lane: CH x 4 pixels (CH x 1 through the scalar walk)
#endif
#define PROJ_AW (PROJ_TW + 1)
#define PROJ_AH (PROJ_TH + 1)
#define PROJ_VS 66                                  // grid row pitch of the 8-byte planes: rows start 16-byte aligned
#define PROJ_CS4 68                                 // ... of the 4-byte count plane
#define PROJ_NW (PROJ_PULL_THREADS / 64)            // waves per workgroup
#ifndef PROJ_ST_AUX
#define PROJ_ST_AUX 2                               // cache policy of K1's result stores: nt (0 = default, 16 = sc1: K1 1.4 us slower at 1080p)
#endif
#define PROJ_EPI_ITERS (PROJ_TH / (4 * PROJ_NW))    // a wave writes four rows per instruction (16 lanes x 16 bytes each)
template <bool DEPTH> struct ProjLds {
    static constexpr int cs = DEPTH ? PROJ_VS : PROJ_CS4;                      // count plane pitch
    static constexpr int accc_off = PROJ_AH * PROJ_VS * 8;
    static constexpr int acc_bytes = accc_off + PROJ_AH * cs * (DEPTH ? 8 : 4);
    static constexpr int total = acc_bytes + PROJ_TH * 8 + 16;                 // + row bitmaps + three counters
};

// The fallback of K1 (a block of K0 reached too many tiles): the reference's own scheme -- this tile as SOURCE tile,
// global atomics into the dense scratch planes [value x | value y | count][batch][h][w] of the workspace (zero between calls)
template <bool DEPTH, int NW>
__device__ __forceinline__ void pull_fallback(const ProjSrc& src, const ProjPlanes& pl, const ProjGeom& g, float* __restrict__ planes,
                                              int b, int ox0, int oy0, int lane, int wave) {
    const unsigned wbits = (unsigned)__float_as_int((float)(g.w - 1)), hbits = (unsigned)__float_as_int((float)(g.h - 1));
    const int per_img = g.tiles_x * g.tiles_y;
    const int64_t npx = (int64_t)(g.ntiles / per_img) * g.h * g.w;
    float* o0 = planes + (int64_t)b * g.h * g.w;
    float* o1 = o0 + npx;
    float* cn = o1 + npx;
    const int x = ox0 + lane;
    for (int r = 0; r < PROJ_TH / NW; ++r) {
        const int y = oy0 + wave * (PROJ_TH / NW) + r;
        if (x >= g.w || y >= g.h) continue;
        const ProjPix p = pix_load<DEPTH>(src, pl, x, 0, y);
        const float fx = p.fx, fy = p.fy;
        int L, T;
        if (!pix_target(fx, fy, x, y, wbits, hbits, L, T)) continue;
        const int R = min(L + 1, g.w - 1), Bm = min(T + 1, g.h - 1);
        const float ax = DEPTH ? -p.d * fx : -fx, ay = DEPTH ? -p.d * fy : -fy, ac = p.d;   // (:75-88; depth :74-91)
        const int64_t oT = (int64_t)T * g.w, oB = (int64_t)Bm * g.w;
        atomicAdd(&o0[oT + L], ax); atomicAdd(&o0[oT + R], ax); atomicAdd(&o0[oB + L], ax); atomicAdd(&o0[oB + R], ax);
        atomicAdd(&o1[oT + L], ay); atomicAdd(&o1[oT + R], ay); atomicAdd(&o1[oB + L], ay); atomicAdd(&o1[oB + R], ay);
        atomicAdd(&cn[oT + L], ac); atomicAdd(&cn[oT + R], ac); atomicAdd(&cn[oB + L], ac); atomicAdd(&cn[oB + R], ac);
    }
}

// The tail of K1, after the barrier behind the tile's stores: the tile's rows of the two "count != 0" bitmaps (from the
// row words collected in LDS) and its word for K2.
template <int THREADS>
__device__ __forceinline__ void pull_bitmaps(const ProjGeom& g, int* __restrict__ ws, int* __restrict__ bits, const unsigned* s_rowbits,
                                             const int* s_misc, int b, int txi, int tyi, int ox0, int oy0, int tile, int tid) {
    // the row-packed bitmap: two words per tile row
    if (tid < 2 * PROJ_TH) {
        const int yl = tid >> 1, wi = txi * 2 + (tid & 1);
        if (oy0 + yl < g.h && wi < g.rmw) bits[g.rowmap + (b * g.h + oy0 + yl) * g.rmw + wi] = (int)s_rowbits[tid];
    }
    // the column-packed one: a column word holds two tiles' rows, each tile stores its own 16-bit half (and the
    // unused halves that pad a column line to whole 16-byte groups, if it is the last tile of the column)
    for (int t = tid; t < PROJ_TW; t += THREADS)
        if (ox0 + t < g.w) {
            unsigned colbits = 0u;
#pragma unroll
            for (int r = 0; r < PROJ_TH; ++r) colbits |= ((s_rowbits[r * 2 + (t >> 5)] >> (t & 31)) & 1u) << r;
            unsigned short* half = reinterpret_cast<unsigned short*>(bits + g.colmap) + ((int64_t)(b * g.w + ox0 + t) * g.cmw) * 2;
            half[tyi] = (unsigned short)colbits;
            if (tyi == g.tiles_y - 1)
                for (int k = tyi + 1; k < 2 * g.cmw; ++k) half[k] = 0;
        }
    // the tile's word for K2: 0 = no holes, 1 = holes, 3 = holes and negative counts (non-zero, yet holes: K2 then
    // reads the counts).  (A list of the tiles with holes, appended with one atomic per tile, costs ~12 ns per tile
    // on its counter -- measured: 24 us more on a rough field where most tiles have holes.)
    if (tid == 0) ws[g.off_list + tile] = s_misc[1] ? (s_misc[2] ? 3 : 1) : 0;
}

// one source pixel into the grid
// cls / ncls / emax: with depth, only the sources of weight class cls are taken (see proj_pull)
template <bool DEPTH>
__device__ __forceinline__ void pull_add(unsigned long long* accv, typename ProjCountCell<DEPTH>::type* accc,
                                         float fx, float fy, float d, int px, int py, bool on, unsigned wbits, unsigned hbits,
                                         int cx, int cy, float svx, float svy, float scn, int cls, int ncls, int emax) {
    int L, T;
    bool valid = pix_target(fx, fy, px, py, wbits, hbits, L, T);
    if constexpr (DEPTH) {
        const int delta = emax - ((__float_as_int(d) >> 23) & 0xff);
        const int mine = min((delta >= PROJ_CLS_BITS ? 1 : 0) + (delta >= 2 * PROJ_CLS_BITS ? 1 : 0) + (delta >= 3 * PROJ_CLS_BITS ? 1 : 0), ncls - 1);
        valid = valid && mine == cls;
    }
    const unsigned c = (unsigned)(L - cx), r = (unsigned)(T - cy);
#if defined(PROJ_STAMPS) && PROJ_DEV_SKIP == 3
    if (valid && on && c < PROJ_AW && r < PROJ_AH && svx == 12345.0f) {
#else
    if (valid && on && c < PROJ_AW && r < PROJ_AH) {
#endif
        // addend * 2^k is exact in float (power-of-two scale)
        const float ax = DEPTH ? d * fx : fx, ay = DEPTH ? d * fy : fy;        // (:75-88; depth :74-91)
        atomicAdd(&accv[r * PROJ_VS + c], pack2(__float2int_rn(ax * svx), __float2int_rn(ay * svy)));
        if constexpr (DEPTH) atomicAdd(&accc[r * PROJ_VS + c], pack2(1, __float2int_rn(d * scn)));
        else atomicAdd(&accc[r * PROJ_CS4 + c], 1u);
    }
}

// VEC: the flow (and depth) rows are 16-byte aligned, so a lane can load four pixels at once
template <bool DEPTH, bool VEC>
__global__ __launch_bounds__(PROJ_PULL_THREADS, PROJ_PULL_WAVES) void proj_pull(
    ProjSrc src, ProjDst dst, ProjGeom g, int64_t ob, int64_t oc, int oh,
    int64_t cb, int ch, int vec_ok, int* __restrict__ ws, int* __restrict__ bits, float* __restrict__ planes, int64_t plane_floats) {
    typedef typename ProjCountCell<DEPTH>::type ccell;
    __shared__ uint4 lds[ProjLds<DEPTH>::total / 16];
    unsigned long long* accv = reinterpret_cast<unsigned long long*>(lds);
    ccell* accc = reinterpret_cast<ccell*>(reinterpret_cast<char*>(lds) + ProjLds<DEPTH>::accc_off);
    unsigned* s_rowbits = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + ProjLds<DEPTH>::acc_bytes);   // [16][2]
    int* s_misc = reinterpret_cast<int*>(s_rowbits + 2 * PROJ_TH);     // [0] most addends in a cell, [1] holes, [2] negative counts
    const int tile = band_item(blockIdx.x, gridDim.x);
    const int per_img = g.tiles_x * g.tiles_y;
    const int b = tile / per_img;
    const int trem = tile - b * per_img;
    const int tyi = trem / g.tiles_x, txi = trem - tyi * g.tiles_x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // (provably wave-uniform: scalar row offsets)
    const int ox0 = txi * PROJ_TW, oy0 = tyi * PROJ_TH;
    const bool fallback = ws[PROJ_WS_FLAG] != 0;
#ifdef PROJ_STAMPS          // development build only: when and where each workgroup ran (tools/proj_stamps.py)
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_t1 = 0, st_t2 = 0;
#endif
    if (tile == 0 && tid == 0) {
        ws[PROJ_WS_DIRTY] = fallback ? (int)(unsigned)(plane_floats & 0xffffffffll) : 0;
        ws[PROJ_WS_DIRTY + 1] = fallback ? (int)(plane_floats >> 32) : 0;
    }
    int* entry = ws + g.off_tile + (int64_t)tile * PROJ_TILE_WORDS;
    const int e0 = entry[0], e1 = entry[1], e2 = entry[2], e3 = entry[3], e4 = entry[4], e5 = entry[5], e6 = entry[6], e7 = entry[7];
    const ProjPlanes pl = proj_planes<DEPTH>(src, b, g.h, g.w);
    const unsigned wbits = (unsigned)__float_as_int((float)(g.w - 1)), hbits = (unsigned)__float_as_int((float)(g.h - 1));

    if (fallback) {
        __syncthreads();
        if (tid < 8) entry[tid] = 0;
        pull_fallback<DEPTH, PROJ_NW>(src, pl, g, planes, b, ox0, oy0, lane, wave);
        return;
    }

    for (int i = tid; i < ProjLds<DEPTH>::total / 16; i += PROJ_PULL_THREADS) lds[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    if (tid < 8) entry[tid] = 0;                            // every thread has read the record: empty for the next call
#ifdef PROJ_STAMPS
    const unsigned long long st_ta = __builtin_amdgcn_s_memtime() + (e0 & 0);      // (after the record has arrived)
#endif

    const int ux0 = 32767 - e0, uy0 = 32767 - e1;
    const int uw = e2 - ux0, uh = e2 > 0 ? e3 - uy0 : 0;    // uh == 0: nothing lands here
    // Fixed-point scales: a value addend is below 2^(ef + ec) and a weight below 2^ec, with 2^ef / 2^ec above the
    // largest |fx| or |fy| (a scale per component: proj_pull_lean) / |weight| that can reach this tile (from K0), so
    // addend * 2^(25 - e) is below 2^25.
    // DepthFlowProjection: the weights (inverse depth, 1e-6 + exp(-d): DAIN_slowmotion.py:143) can span many
    // orders of magnitude inside one tile, e.g. at the edge of a near object in front of sky, and a cell that only
    // far-away sources reach must still get full relative precision (the reference's fp32 sums give it that).  So
    // the sources are taken in up to four passes by weight class -- class j: weights within 2^(-6 j) .. 2^(-6 j - 6)
    // of the largest, the last class everything below -- each pass with its own scale, exact integer sums, one
    // rounding to float, and the classes' results are added.  Almost every tile has one class.
    int efx = 0, efy = 0, ec = 0;
    (void)frexpf(__int_as_float(e4), &efx);
    (void)frexpf(__int_as_float(e7), &efy);
    if constexpr (DEPTH) (void)frexpf(__int_as_float(e5), &ec);
    const int emax = (e5 >> 23) & 0xff, emin = e6 ? ((PROJ_INV_BITS - e6) >> 23) & 0xff : emax;
    const int ncls = DEPTH ? min(4, max(0, emax - emin) / PROJ_CLS_BITS + 1) : 1;
    // epilogue geometry: a lane owns four consecutive cells of a row, 16 lanes a row, a wave four rows
    const int q = lane & 15, rw = lane >> 4;
    const int xq = ox0 + 4 * q;
    float resx[PROJ_EPI_ITERS][4], resy[PROJ_EPI_ITERS][4], resc[PROJ_EPI_ITERS][4];    // the tile's sums, not yet normalised
    // the count cells a lane's four tile cells are made of, summed (the number of addends sits in the high half with depth)
    auto count_sums = [&](int it, ccell (&c4)[4]) {
        const int yl = it * 4 * PROJ_NW + wave * 4 + rw;
        ccell n[2][5];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if constexpr (DEPTH) {
                const uint4* pc = reinterpret_cast<const uint4*>(accc + (yl + r) * PROJ_VS + 4 * q);
                const uint4 c01 = pc[0], c23 = pc[1];
                n[r][0] = c01.x | ((unsigned long long)c01.y << 32); n[r][1] = c01.z | ((unsigned long long)c01.w << 32);
                n[r][2] = c23.x | ((unsigned long long)c23.y << 32); n[r][3] = c23.z | ((unsigned long long)c23.w << 32);
                n[r][4] = accc[(yl + r) * PROJ_VS + 4 * q + 4];
            } else {
                const uint4 c03 = *reinterpret_cast<const uint4*>(accc + (yl + r) * PROJ_CS4 + 4 * q);
                n[r][0] = c03.x; n[r][1] = c03.y; n[r][2] = c03.z; n[r][3] = c03.w;
                n[r][4] = accc[(yl + r) * PROJ_CS4 + 4 * q + 4];
            }
        }
        const bool ylast = oy0 + yl == g.h - 1;                            // B == T there: the row adds twice
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool xlast = xq + j == g.w - 1;                          // R == L there: the column adds twice
            ccell c0 = n[0][j] + n[0][j + 1], c1 = n[1][j] + n[1][j + 1];
            if (xlast) { c0 += n[0][j + 1]; c1 += n[1][j + 1]; }
            ccell c = c0 + c1;
            if (ylast) c += c1;
            c4[j] = c;
        }
    };
    // one weight class: accumulate, check, sum.  FIRST: the results are assigned, not added -- the one-class case, almost
    // every tile, then holds no partial results in registers across the accumulation loop (two instantiations)
    auto run_class = [&](auto first_tag, int cls) {
    constexpr bool FIRST = decltype(first_tag)::value;
    // (clamped so that 2^k stays a normal float when every addend is tiny or huge)
    int kc = max(-100, min(100, PROJ_ADD_BITS - (ec - PROJ_CLS_BITS * cls)));
    int kvx = max(-100, min(100, PROJ_ADD_BITS - (efx + ec - PROJ_CLS_BITS * cls)));
    int kvy = max(-100, min(100, PROJ_ADD_BITS - (efy + ec - PROJ_CLS_BITS * cls)));
    for (int attempt = 0;; ++attempt) {
        const float svx = -ldexpf(1.0f, kvx), svy = -ldexpf(1.0f, kvy), scn = ldexpf(1.0f, kc);    // exact powers of two (the value addend is MINUS the flow)
        // The rectangle is walked in strips of up to 64 lanes.  A full strip: the workgroup's waves take its rows
        // in turn (row base and flow row pointer advance on the scalar unit: no per-pixel index arithmetic).  A
        // narrower strip packs 64 / width rows into a wave the same way.  Rows past the rectangle are not loaded.
        constexpr int CH = PROJ_PULL_CH;
        if constexpr (!VEC) {
            for (int cs = 0; cs < uw && uh > 0; cs += 64) {
                const int width = min(64, uw - cs), rpi = 64 / width;      // rows per wave instruction
                const int lr = lane / width, lc = lane - lr * width;
                const int px = ux0 + cs + lc;
                const bool lane_on = lr < rpi;
                const int step = PROJ_NW * rpi;
                for (int row0 = wave * rpi; row0 < uh; row0 += CH * step) {
                    ProjPix raw[CH];
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        raw[k].fx = raw[k].fy = raw[k].d = 0.0f;
                        if (row0 + k * step < uh) raw[k] = pix_load<DEPTH>(src, pl, px, lr, uy0 + row0 + k * step);    // (past the plane: reads 0)
                    }
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int rowk = row0 + k * step + lr;
                        const int py = uy0 + rowk;
                        pull_add<DEPTH>(accv, accc, raw[k].fx, raw[k].fy, raw[k].d, px, py, lane_on && rowk < uh, wbits, hbits, ox0 - 1, oy0 - 1, svx, svy, scn,
                                        cls, ncls, emax);
                    }
                }
            }
        } else {
            const int ux0a = ux0 & ~3;                                      // quads start at multiples of four pixels
#if defined(PROJ_STAMPS) && PROJ_DEV_SKIP == 1
            const int nq = 0;
#else
            const int nq = uh > 0 ? (ux0 + uw - 1 - ux0a) / 4 + 1 : 0;      // quads per row
#endif
            for (int cs = 0; cs < nq; cs += 64) {
                const int width = min(64, nq - cs), rpi = 64 / width;
                const int lr = lane / width, lc = lane - lr * width;
                const int px = ux0a + 4 * (cs + lc);
                const bool lane_on = lr < rpi;
                const int step = PROJ_NW * rpi;
                const int vo = (lr * src.fh + px) * 4, vod = (lr * src.dh + px) * 4;
                for (int row0 = wave * rpi; row0 < uh; row0 += CH * step) {
                    proj_v4f qx[CH], qy[CH], qd[CH];
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        qx[k] = qy[k] = qd[k] = proj_v4f{0.0f, 0.0f, 0.0f, 0.0f};
                        if (row0 + k * step < uh) {                         // (wave-uniform test)
                            const int yu = uy0 + row0 + k * step;
                            qx[k] = buf_f32x4(pl.f0, vo, yu * src.fh * 4);
                            qy[k] = buf_f32x4(pl.f1, vo, yu * src.fh * 4);
                            if constexpr (DEPTH) qd[k] = buf_f32x4(pl.d, vod, yu * src.dh * 4);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int rowk = row0 + k * step + lr;
                        const bool on = lane_on && rowk < uh;
#pragma unroll
                        for (int j = 0; j < 4; ++j)                         // (a quad may reach past the row: x < w)
                            pull_add<DEPTH>(accv, accc, qx[k][j], qy[k][j], qd[k][j], px + j, uy0 + rowk, on && px + j < g.w, wbits, hbits,
                                            ox0 - 1, oy0 - 1, svx, svy, scn, cls, ncls, emax);
                    }
                }
            }
        }
#ifdef PROJ_STAMPS
        if (!attempt) st_t1 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
        if (attempt) break;
        // did every cell stay within the addends its 32-bit halves can hold?
        int nmax = 0;
#pragma unroll
        for (int it = 0; it < PROJ_EPI_ITERS; ++it) {
            ccell c4[4];
            count_sums(it, c4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (xq + j < g.w && oy0 + it * 4 * PROJ_NW + wave * 4 + rw < g.h) {
                    if constexpr (DEPTH) nmax = max(nmax, packed_hi(c4[j])); else nmax = max(nmax, (int)min(c4[j], (ccell)0x7fffffffu));
                }
        }
        nmax = wave_max_i32(nmax);
        if (lane == 0 && nmax > PROJ_ADD_CELL) atomicMax(&s_misc[0], nmax);
        __syncthreads();
        nmax = s_misc[0];
        if (nmax <= PROJ_ADD_CELL) break;
        // once more with addends small enough for the busiest cell
        const int shift = (32 - __clz(nmax - 1)) - 5;           // ceil(log2(nmax)) - log2(32)
        kvx -= shift; kvy -= shift; kc -= shift;
        for (int i = tid; i < ProjLds<DEPTH>::acc_bytes / 16; i += PROJ_PULL_THREADS) lds[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    // a cell = the grid cells of the top-left targets (x, y), (x - 1, y), (x, y - 1), (x - 1, y - 1): a lane reads the
    // five grid cells above and the five beside its four cells; exact integer sums -> float once (per class)
#pragma unroll
    for (int it = 0; it < PROJ_EPI_ITERS; ++it) {
        const int yl = it * 4 * PROJ_NW + wave * 4 + rw;
        unsigned long long a[2][5];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint4* pv = reinterpret_cast<const uint4*>(accv + (yl + r) * PROJ_VS + 4 * q);
            const uint4 v01 = pv[0], v23 = pv[1];
            a[r][0] = v01.x | ((unsigned long long)v01.y << 32); a[r][1] = v01.z | ((unsigned long long)v01.w << 32);
            a[r][2] = v23.x | ((unsigned long long)v23.y << 32); a[r][3] = v23.z | ((unsigned long long)v23.w << 32);
            a[r][4] = accv[(yl + r) * PROJ_VS + 4 * q + 4];
        }
        ccell c4[4];
        count_sums(it, c4);
        const bool ylast = oy0 + yl == g.h - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool xlast = xq + j == g.w - 1;
            unsigned long long h0 = a[0][j] + a[0][j + 1], h1 = a[1][j] + a[1][j + 1];
            if (xlast) { h0 += a[0][j + 1]; h1 += a[1][j + 1]; }
            unsigned long long v = h0 + h1;
            if (ylast) v += h1;
            const float px_ = ldexpf((float)packed_hi(v), -kvx), py_ = ldexpf((float)packed_lo(v), -kvy);
            float pc_;
            if constexpr (DEPTH) pc_ = ldexpf((float)packed_lo(c4[j]), -kc); else pc_ = (float)c4[j];
            if constexpr (FIRST) { resx[it][j] = px_; resy[it][j] = py_; resc[it][j] = pc_; }
            else { resx[it][j] += px_; resy[it][j] += py_; resc[it][j] += pc_; }
        }
    }
    };
    run_class(std::true_type{}, 0);
    for (int cls = 1; cls < ncls; ++cls) {
        __syncthreads();                                        // every lane has read the previous class's sums
        for (int i = tid; i < ProjLds<DEPTH>::acc_bytes / 16; i += PROJ_PULL_THREADS) lds[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        run_class(std::false_type{}, cls);
    }
#ifdef PROJ_STAMPS
    st_t2 = __builtin_amdgcn_s_memtime();
#endif

    // normalise (flowprojection_cuda_kernel.cu:129-134) and write the tile once, 16 bytes per lane; leave the two
    // "count != 0" bitmaps for the hole filler and put the tile on its list if it has holes
    int holes = 0, negs = 0;
    const ProjImage im = proj_image(b, src.per);
    float* const out = dst.out[im.item] + (int64_t)im.bi * ob;
    float* const count = dst.count[im.item] + (int64_t)im.bi * cb;
    const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, ((g.h - 1) * oh + g.w) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc((void*)(out + oc), 0, ((g.h - 1) * oh + g.w) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcn = __builtin_amdgcn_make_buffer_rsrc((void*)count, 0, ((g.h - 1) * ch + g.w) * 4, 0x00020000);
#pragma unroll
    for (int it = 0; it < PROJ_EPI_ITERS; ++it) {
        const int yl = it * 4 * PROJ_NW + wave * 4 + rw, y = oy0 + yl;
        float vxa[4], vya[4], ca[4];
        unsigned nzb = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool inside = xq + j < g.w && y < g.h;
            const float c = resc[it][j];
            float vx = resx[it][j], vy = resy[it][j];
            if (c > 0.0f) { vx /= c; vy /= c; }
            vxa[j] = vx; vya[j] = vy; ca[j] = c;
            if (inside && c != 0.0f) nzb |= 1u << j;
            if (inside && c <= 0.0f) holes += 1;
            if (DEPTH && inside && c < 0.0f) negs += 1;
        }
        if (nzb) atomicOr(&s_rowbits[yl * 2 + (q >> 3)], nzb << ((q & 7) * 4));
#if defined(PROJ_STAMPS) && PROJ_DEV_SKIP == 2
        if (y < g.h && ncls == 12345) {
#else
        if (y < g.h) {
#endif
            const int so0 = (oy0 + it * 4 * PROJ_NW + wave * 4) * oh * 4, soc = (oy0 + it * 4 * PROJ_NW + wave * 4) * ch * 4;
            const int vo = (rw * oh + xq) * 4, voc = (rw * ch + xq) * 4;
            if (vec_ok && xq + 3 < g.w) {
                const proj_v4i vx4 = {__float_as_int(vxa[0]), __float_as_int(vxa[1]), __float_as_int(vxa[2]), __float_as_int(vxa[3])};
                const proj_v4i vy4 = {__float_as_int(vya[0]), __float_as_int(vya[1]), __float_as_int(vya[2]), __float_as_int(vya[3])};
                const proj_v4i c4 = {__float_as_int(ca[0]), __float_as_int(ca[1]), __float_as_int(ca[2]), __float_as_int(ca[3])};
                __builtin_amdgcn_raw_buffer_store_b128(vx4, ro0, vo, so0, PROJ_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(vy4, ro1, vo, so0, PROJ_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(c4, rcn, voc, soc, PROJ_ST_AUX);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (xq + j < g.w) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(vxa[j]), ro0, vo + 4 * j, so0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(vya[j]), ro1, vo + 4 * j, so0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ca[j]), rcn, voc + 4 * j, soc, 0);
                    }
            }
        }
    }
    if (holes) atomicAdd(&s_misc[1], holes);
    if (negs) atomicAdd(&s_misc[2], negs);
    __syncthreads();
    pull_bitmaps<PROJ_PULL_THREADS>(g, ws, bits, s_rowbits, s_misc, b, txi, tyi, ox0, oy0, tile, tid);
#ifdef PROJ_STAMPS
    if (tid == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(ws + g.off_list + g.ntiles + (g.ntiles & 1)) + (int64_t)tile * 8;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st[0] = st_t0; st[1] = st_t1; st[2] = st_t2; st[3] = __builtin_amdgcn_s_memtime();
        st[4] = st_r0; st[5] = __builtin_amdgcn_s_memrealtime(); st[6] = hwid; st[7] = (unsigned long long)xcc | ((st_ta - st_t0) << 8);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// K1, lean (round 3): full-resolution flow with 16-byte aligned rows -- what FlowProjection / DepthFlowProjection are
// called with.  Same tile and same grid of exact integer sums as proj_pull, written for the fewest instructions per
// pixel, because that is what bounds the kernel: all 2232 tiles of a 1080p frame are resident at once, 4.5 waves per
// SIMD, and a wave of proj_pull executes ~2000 vector instructions for its ~11 source and 8 output pixels per lane
// (SQ_INSTS_VALU; the workgroups dispatched first finish first -- age-ordered issue -- which is what the in-kernel
// stamps show: tools/proj_stamps.py).  Here:
//  * a source pixel is tested ONCE: "valid target" (0 <= x2 <= w - 1) and "target inside this tile's grid" are one
//    interval per axis, tested on the float's bits (non-negative floats order like their bit patterns: one subtract, one
//    unsigned compare).  No lane predicates beside it: lanes past the frame or past the strip carry NaN coordinates,
//    and an in-frame pixel outside the rectangle cannot reach the tile (K0's rectangle holds every pixel that can);
//  * grid cell address = (T * pitch + L) * 8 + one uniform constant; the count plane has the value plane's 8-byte
//    pitch, so its address is the same register + an immediate;
//  * the y half of a value addend is biased by 2^25 (non-negative: the 64-bit add never borrows from the x half; the
//    epilogue subtracts count * 2^25), which saves the sign fix-up per pixel and the packed-half recovery per cell;
//  * the 2x2 sums of a lane's four cells share their five column sums; the frame's last row / column (R == L, B == T:
//    the reference adds twice) is fixed up in the tiles that touch them only;
//  * FlowProjection divides by an integer count <= 32 times a power of two: the reciprocal is rcp + one Newton step
//    (= the correctly rounded 1 / n for every n <= 32: tests), the quotient one multiply and one correction per
//    component (Markstein: correctly rounded, the same bits as an IEEE division);
//  * the record comes through the scalar cache (one s_load_dwordx8), tile coordinates from a 3-D grid (no divisions);
//  * the busiest-cell check (more than 32 addends in a cell: accumulate again with a coarser scale) rides on the
//    epilogue's own count sums and on the barrier before the bitmaps instead of a pass and a barrier of its own: the
//    tile is written as if every cell were fine, and written again in the rare case one was not.
// LDS layout.  gfx950 allots LDS in granules of 1280 bytes and nine workgroups per CU (all 2232 tiles of a 1080p frame resident
// at once, no second round) may use 14 granules each = 17,920 bytes.  FlowProjection: 8-byte value cells at a pitch of 66 (rows
// 16-byte aligned: ds_read_b128) + 4-byte count cells at a pitch of 68 = 13,744 bytes.  DepthFlowProjection needs 8-byte count
// cells ({weight sum, addends}): with both planes at a pitch of 65 (rows 8-byte aligned: ds_read2_b64) it would fit in 17,824 bytes,
// but its 127 registers allow eight workgroups per CU anyway (at 96, nine per CU, it spills and takes 30.8 instead of 25 us
// at 1080p): pitch 66, 18,096 bytes, a short second round of 184 tiles.
template <bool DEPTH> struct PlLds {
    static constexpr int P = 66;                                               // value plane pitch, cells
    static constexpr int PC = DEPTH ? 66 : 68;                                 // count plane pitch, cells
    static constexpr int CB = DEPTH ? 8 : 4;                                   // bytes per count cell
    static constexpr int vplane = (PROJ_AH * P * 8 + 15) & ~15;
    static constexpr int cplane = (PROJ_AH * PC * CB + 15) & ~15;
    static constexpr int planes = vplane + cplane;
    static constexpr int total = planes + PROJ_TH * 8 + 16;                    // + row bitmaps + four counters
};
static_assert(PlLds<false>::total <= 17920, "nine workgroups per CU");
#define PL_BIAS (1 << PROJ_ADD_BITS)
#ifndef PL_THREADS
#define PL_THREADS 128
#endif
#ifndef PL_CH
#define PL_CH 4                                     // row steps whose loads are in flight together (depth: half, for the registers)
#endif
#define PL_NW (PL_THREADS / 64)
#define PL_EPI (PROJ_TH / (4 * PL_NW))              // a wave writes four rows per pass

typedef int pl_v8i __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned long long* pl_lds64_t;
typedef __attribute__((address_space(3))) unsigned* pl_lds32_t;

// a / d for an integer |a| < 2^31 (converted exactly, or rounded once) and d = n * 2^k, n a positive integer: with r the
// correctly rounded 1 / d, q0 = a * r is within an ulp of a / d, the remainder is exact and the corrected quotient is
// RN(a / d) (Markstein's theorem) -- the bits of an IEEE division.  (k keeps everything in the normal range: the host
// path clamps it.)
__device__ __forceinline__ float pl_div(float a, float d, float r) {
    const float q0 = a * r;
    const float rem = fmaf(-d, q0, a);
    return fmaf(rem, r, q0);
}
__device__ __forceinline__ float pl_rcp(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, r0, 1.0f);
    return fmaf(e, r0, r0);
}

template <bool DEPTH>
__global__ __launch_bounds__(PL_THREADS, DEPTH ? 4 : 5) void proj_pull_lean(
    ProjSrc src, ProjDst dst, ProjGeom g, int64_t ob, int64_t oc, int oh,
    int64_t cb, int ch, int* __restrict__ ws, int* __restrict__ bits, float* __restrict__ planes, int64_t plane_floats) {
    typedef PlLds<DEPTH> LY;
    constexpr int P = LY::P, PC = LY::PC;
    __shared__ uint4 lds[LY::total / 16];
    char* const lbase = reinterpret_cast<char*>(lds);
    unsigned* s_rowbits = reinterpret_cast<unsigned*>(lbase + LY::planes);            // [16][2]
    int* s_misc = reinterpret_cast<int*>(s_rowbits + 2 * PROJ_TH);                    // [0] most addends in a cell, [1] holes, [2] negative counts
#if PROJ_BANDS
    const int tile = band_item(blockIdx.x, gridDim.x);
    const int per_img = g.tiles_x * g.tiles_y;
    const int b = tile / per_img;
    const int trem = tile - b * per_img;
    const int tyi = trem / g.tiles_x, txi = trem - tyi * g.tiles_x;
#else
    const int txi = blockIdx.x, tyi = blockIdx.y, b = blockIdx.z;
    const int tile = (b * g.tiles_y + tyi) * g.tiles_x + txi;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ox0 = txi * PROJ_TW, oy0 = tyi * PROJ_TH;
    int* entry = ws + g.off_tile + (int64_t)tile * PROJ_TILE_WORDS;
    // the record and the fallback word through the scalar cache (invalidated at kernel start; K0 wrote them by atomics)
#ifdef PROJ_STAMPS          // development build only: when and where each workgroup ran (tools/proj_stamps.py)
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_t1 = 0, st_t2 = 0;
#endif
    pl_v8i rec;
    int fb_word;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0" : "=&s"(rec), "=&s"(fb_word) : "s"(entry), "s"(ws + PROJ_WS_FLAG) : "memory");
    for (int i = tid; i < LY::total / 16; i += PL_THREADS) lds[i] = make_uint4(0u, 0u, 0u, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rec), "+s"(fb_word) :: "memory");
    const bool fallback = fb_word != 0;
#ifdef PROJ_STAMPS
    const unsigned long long st_ta = __builtin_amdgcn_s_memtime();             // (after the record has arrived)
#endif
    if (tile == 0 && tid == 0) {
        ws[PROJ_WS_DIRTY] = fallback ? (int)(unsigned)(plane_floats & 0xffffffffll) : 0;
        ws[PROJ_WS_DIRTY + 1] = fallback ? (int)(plane_floats >> 32) : 0;
    }
    const ProjPlanes pl = proj_planes<DEPTH>(src, b, g.h, g.w);
    __syncthreads();
    if (tid < 8) entry[tid] = 0;                            // the record is in registers: empty for the next call
    if (fallback) {
        pull_fallback<DEPTH, PL_NW>(src, pl, g, planes, b, ox0, oy0, lane, wave);
        return;
    }

    const int e0 = rec[0], e1 = rec[1], e2 = rec[2], e3 = rec[3], e4 = rec[4], e5 = rec[5], e6 = rec[6], e7 = rec[7];
    const int ux0 = 32767 - e0, uy0 = 32767 - e1;
    const int uw = e2 - ux0, uh = e2 > 0 ? e3 - uy0 : 0;    // uh == 0: nothing lands here
    // Fixed-point scales (see proj_pull for the weight classes).  Each flow component has its own: a tile's x scale follows the
    // largest |fx| that reaches it, its y scale the largest |fy|, so a component that is small all over the tile's sources --
    // the zero crossings of a smooth field, a pan along one axis -- keeps its own precision beside a large other component
    // (one common scale left 1.6e-4 px on such cells under 256-pixel flows; SURVEY's tolerance is 1e-4).
    int efx = 0, efy = 0, ec = 0;
    (void)frexpf(__int_as_float(e4), &efx);
    (void)frexpf(__int_as_float(e7), &efy);
    if constexpr (DEPTH) (void)frexpf(__int_as_float(e5), &ec);
    const int emax = (e5 >> 23) & 0xff, emin = e6 ? ((PROJ_INV_BITS - e6) >> 23) & 0xff : emax;
    const int ncls = DEPTH ? min(4, max(0, emax - emin) / PROJ_CLS_BITS + 1) : 1;
    // one interval per axis: valid target and inside the grid (columns ox0 - 1 .. ox0 + 63, rows oy0 - 1 .. oy0 + 15)
    const float lox = (float)max(ox0 - 1, 0), loy = (float)max(oy0 - 1, 0);
    const float hix = fminf((float)(g.w - 1), __uint_as_float(__float_as_uint((float)(ox0 + PROJ_TW)) - 1u));
    const float hiy = fminf((float)(g.h - 1), __uint_as_float(__float_as_uint((float)(oy0 + PROJ_TH)) - 1u));
    const unsigned lxb = __float_as_uint(lox), lyb = __float_as_uint(loy);
    const unsigned spx = __float_as_uint(hix) - lxb, spy = __float_as_uint(hiy) - lyb;
    // byte address of value cell (T, L) = (T * P + L) * 8 + cell0; count cell: (T * PC + L) * CB + cellc (depth: same pitch,
    // same size: the value cell's address + an immediate)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lbase;
    const unsigned cell0 = lds0 - 8u * (unsigned)((oy0 - 1) * P + (ox0 - 1));
    const unsigned cellc = lds0 + LY::vplane - (unsigned)LY::CB * (unsigned)((oy0 - 1) * PC + (ox0 - 1));

    const int q = lane & 15, rw = lane >> 4;
    const int xq = ox0 + 4 * q;
    const bool edge_x = ox0 + PROJ_TW >= g.w, edge_y = oy0 + PROJ_TH >= g.h;     // the tile touches the frame's last column / row
    const ProjImage im = proj_image(b, src.per);
    float* const out = dst.out[im.item] + (int64_t)im.bi * ob;
    float* const count = dst.count[im.item] + (int64_t)im.bi * cb;
    const __amdgpu_buffer_rsrc_t ro0 = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, ((g.h - 1) * oh + g.w) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro1 = __builtin_amdgcn_make_buffer_rsrc((void*)(out + oc), 0, ((g.h - 1) * oh + g.w) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rcn = __builtin_amdgcn_make_buffer_rsrc((void*)count, 0, ((g.h - 1) * ch + g.w) * 4, 0x00020000);

    // the 2x2 sums of one pass for the lane's four cells: lo / hi halves of the value cells; count cells: clo = weight sum
    // (depth), chi = addends
    auto sums4 = [&](int it, unsigned (&vlo)[4], int (&vhi)[4], unsigned (&clo)[4], int (&chi)[4]) {
        const int yl = it * 4 * PL_NW + wave * 4 + rw;
        unsigned l0[5], l1[5], h0[5], h1[5], m0[5], m1[5], n0[5], n1[5];
        auto row_v = [&](int r, unsigned (&lo)[5], unsigned (&hi)[5]) {
            const char* pr = lbase + ((yl + r) * P + 4 * q) * 8;
            const uint4 a = *reinterpret_cast<const uint4*>(pr), c = *reinterpret_cast<const uint4*>(pr + 16);
            const uint2 e = *reinterpret_cast<const uint2*>(pr + 32);
            lo[0] = a.x; hi[0] = a.y; lo[1] = a.z; hi[1] = a.w; lo[2] = c.x; hi[2] = c.y; lo[3] = c.z; hi[3] = c.w; lo[4] = e.x; hi[4] = e.y;
        };
        auto row_c = [&](int r, unsigned (&w)[5], unsigned (&n)[5]) {
            if constexpr (DEPTH) {
                const char* pr = lbase + LY::vplane + ((yl + r) * PC + 4 * q) * 8;
                const uint4 a = *reinterpret_cast<const uint4*>(pr), c = *reinterpret_cast<const uint4*>(pr + 16);
                const uint2 e = *reinterpret_cast<const uint2*>(pr + 32);
                w[0] = a.x; n[0] = a.y; w[1] = a.z; n[1] = a.w; w[2] = c.x; n[2] = c.y; w[3] = c.z; n[3] = c.w; w[4] = e.x; n[4] = e.y;
            } else {
                const char* pr = lbase + LY::vplane + ((yl + r) * PC + 4 * q) * 4;
                const uint4 a = *reinterpret_cast<const uint4*>(pr);
                n[0] = a.x; n[1] = a.y; n[2] = a.z; n[3] = a.w; n[4] = *reinterpret_cast<const unsigned*>(pr + 16);
#pragma unroll
                for (int k = 0; k < 5; ++k) w[k] = 0u;
            }
        };
        row_v(0, l0, h0); row_v(1, l1, h1); row_c(0, m0, n0); row_c(1, m1, n1);
        if (edge_y && oy0 + yl == g.h - 1) {                // B == T: the row below adds twice
#pragma unroll
            for (int k = 0; k < 5; ++k) { l1[k] += l1[k]; h1[k] += h1[k]; m1[k] += m1[k]; n1[k] += n1[k]; }
        }
        unsigned sl[5], sh[5], sm[5], sn[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) { sl[k] = l0[k] + l1[k]; sh[k] = h0[k] + h1[k]; sm[k] = m0[k] + m1[k]; sn[k] = n0[k] + n1[k]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vlo[j] = sl[j] + sl[j + 1]; vhi[j] = (int)(sh[j] + sh[j + 1]);
            clo[j] = sm[j] + sm[j + 1]; chi[j] = (int)(sn[j] + sn[j + 1]);
        }
        if (edge_x) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (xq + j == g.w - 1) { vlo[j] += sl[j + 1]; vhi[j] += (int)sh[j + 1]; clo[j] += sm[j + 1]; chi[j] += (int)sn[j + 1]; }   // R == L
        }
    };

    // one weight class into the grid: every source pixel of the rectangle, once, at its top-left target
    auto accumulate = [&](int cls, int kvx, int kvy, int kc) {
        constexpr int CH = PL_CH;
        const float svx = -ldexpf(1.0f, kvx), svy = -ldexpf(1.0f, kvy), scn = ldexpf(1.0f, kc);    // exact powers of two (the value addend is MINUS the flow)
        const int ux0a = ux0 & ~3;                                          // quads start at multiples of four pixels
#if PROJ_DEV_SKIP == 1      // (development: timing of the kernel without its source walk)
        const int nq = 0;
#else
        const int nq = uh > 0 ? (ux0 + uw - 1 - ux0a) / 4 + 1 : 0;          // quads per row
#endif
        for (int cs = 0; cs < nq; cs += 64) {
            // a strip of up to 64 quads; a narrower one packs 64 / width rows into a wave instruction
            const int width = min(64, nq - cs), rpi = 64 / width;
            const int lr = lane / width, lc = lane - lr * width;
            const int px = ux0a + 4 * (cs + lc);
            const int step = PL_NW * rpi;
            const int vo = (lr * src.fh + px) * 4, vod = (lr * src.dh + px) * 4;
            // lanes beyond the strip's rows and pixels beyond the frame's last column: NaN coordinates, never inside
            float pxf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) pxf[j] = (lr < rpi && px + j < g.w) ? (float)(px + j) : __int_as_float(0x7fc00000);
            float yf = (float)(uy0 + wave * rpi + lr);
            const float ystep = (float)step;
            for (int row0 = wave * rpi; row0 < uh; row0 += CH * step) {
                proj_v4f qx[CH], qy[CH], qd[CH];
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    // (rows of the frame below the rectangle cannot reach the tile, rows below the frame read 0 and sit at
                    //  y > h - 1: both fail the interval test)
                    qx[k] = qy[k] = qd[k] = proj_v4f{0.0f, 0.0f, 0.0f, 0.0f};
                    if (row0 + k * step < uh && PROJ_DEV_SKIP != 4) {       // (wave-uniform)
                        const int yu = uy0 + row0 + k * step;
                        qx[k] = buf_f32x4(pl.f0, vo, yu * src.fh * 4);
                        qy[k] = buf_f32x4(pl.f1, vo, yu * src.fh * 4);
                        if constexpr (DEPTH) qd[k] = buf_f32x4(pl.d, vod, yu * src.dh * 4);
                    }
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (row0 + k * step < uh) {                             // (wave-uniform)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float fx = qx[k][j], fy = qy[k][j];
                            const float x2 = pxf[j] + fx, y2 = yf + fy;                      // (:66-67)
                            bool hit = (__float_as_uint(x2) - lxb) <= spx && (__float_as_uint(y2) - lyb) <= spy;
                            float d = 1.0f;
                            if constexpr (DEPTH) {
                                d = qd[k][j];
                                if (ncls > 1) {
                                    const int delta = emax - ((__float_as_int(d) >> 23) & 0xff);
                                    const int mine = min((delta >= PROJ_CLS_BITS ? 1 : 0) + (delta >= 2 * PROJ_CLS_BITS ? 1 : 0) + (delta >= 3 * PROJ_CLS_BITS ? 1 : 0), ncls - 1);
                                    hit = hit && mine == cls;
                                }
                            }
#if PROJ_DEV_SKIP == 3
                            hit = hit && svx == 12345.0f;
#endif
                            if (hit) {
                                const int L = (int)x2, T = (int)y2;
                                const unsigned a = (unsigned)(__mul24(T, P) + L) * 8u + cell0;
                                const float ax = DEPTH ? d * fx : fx, ay = DEPTH ? d * fy : fy;  // (:75-88; depth :74-91)
                                const unsigned X = (unsigned)__float2int_rn(ax * svx), Y = (unsigned)(__float2int_rn(ay * svy) + PL_BIAS);
                                __hip_atomic_fetch_add((pl_lds64_t)a, ((unsigned long long)X << 32) | Y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if constexpr (DEPTH) {
                                    const unsigned Wc = (unsigned)(__float2int_rn(d * scn) + PL_BIAS);
                                    __hip_atomic_fetch_add((pl_lds64_t)(a + (cellc - cell0)), (1ull << 32) | Wc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                } else {
                                    const unsigned ac = (unsigned)(__mul24(T, PC) + L) * 4u + cellc;
                                    __hip_atomic_fetch_add((pl_lds32_t)ac, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            }
                        }
                    }
                    yf += ystep;
                }
            }
        }
    };

    int shift = 0;                                          // coarser scale of the second try
    for (int attempt = 0;; ++attempt) {
        float resx[PL_EPI][4], resy[PL_EPI][4], resc[PL_EPI][4];
        int nres[PL_EPI][4];
        int nmax = 0;
        for (int cls = 0; cls < ncls; ++cls) {
            if (cls > 0) {
                __syncthreads();                                // every lane has the previous class's sums
                for (int i = tid; i < LY::planes / 16; i += PL_THREADS) lds[i] = make_uint4(0u, 0u, 0u, 0u);
                __syncthreads();
            }
            // (clamped so that 2^k stays a normal float when every addend is tiny or huge)
            const int kc = max(-100, min(100, PROJ_ADD_BITS - (ec - PROJ_CLS_BITS * cls))) - shift;
            const int kvx = max(-100, min(100, PROJ_ADD_BITS - (efx + ec - PROJ_CLS_BITS * cls))) - shift;
            const int kvy = max(-100, min(100, PROJ_ADD_BITS - (efy + ec - PROJ_CLS_BITS * cls))) - shift;
            accumulate(cls, kvx, kvy, kc);
#ifdef PROJ_STAMPS
            if (!attempt && !cls) st_t1 = __builtin_amdgcn_s_memtime();
#endif
            __syncthreads();
#pragma unroll
            for (int it = 0; it < PL_EPI; ++it) {
                unsigned vlo[4], clo[4];
                int vhi[4], chi[4];
                sums4(it, vlo, vhi, clo, chi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = chi[j];                       // addends of the cell (cells outside the frame: none)
                    nmax = max(nmax, n);
                    const int Y = (int)(vlo[j] - ((unsigned)n << PROJ_ADD_BITS));
                    // exact integer sums -> float once per class (a power-of-two scale: exact)
                    const float px_ = ldexpf((float)vhi[j], -kvx), py_ = ldexpf((float)Y, -kvy);
                    float pc_;
                    if constexpr (DEPTH) pc_ = ldexpf((float)(int)(clo[j] - ((unsigned)n << PROJ_ADD_BITS)), -kc);
                    else pc_ = 0.0f;
                    if (cls == 0) { resx[it][j] = px_; resy[it][j] = py_; resc[it][j] = pc_; nres[it][j] = n; }
                    else { resx[it][j] += px_; resy[it][j] += py_; resc[it][j] += pc_; nres[it][j] += n; }
                }
            }
        }
        if (__builtin_amdgcn_ballot_w64(nmax > PROJ_ADD_CELL) != 0ull) {
            nmax = wave_max_i32(nmax);
            if (lane == 0) atomicMax(&s_misc[0], nmax);
        }

        // normalise (flowprojection_cuda_kernel.cu:129-134), write the tile once, 16 bytes per lane; leave the bitmaps of
        // "count != 0" for the hole filler
        int holes = 0, negs = 0;
#pragma unroll
        for (int it = 0; it < PL_EPI; ++it) {
            const int yl = it * 4 * PL_NW + wave * 4 + rw, y = oy0 + yl;
            float vxa[4], vya[4], ca[4];
            unsigned nzb = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool inside = xq + j < g.w && y < g.h;
                float c, vx, vy;
                if constexpr (DEPTH) {
                    c = resc[it][j]; vx = resx[it][j]; vy = resy[it][j];
                    if (c > 0.0f) { vx /= c; vy /= c; }
                    if (inside && c < 0.0f) negs += 1;
                } else {
                    c = (float)nres[it][j];
                    const float dd = fmaxf(c, 1.0f);            // sums / count; no addends: zero sums
                    const float r = pl_rcp(dd);
                    vx = pl_div(resx[it][j], dd, r); vy = pl_div(resy[it][j], dd, r);
                }
                vxa[j] = vx; vya[j] = vy; ca[j] = c;
                if (inside && c != 0.0f) nzb |= 1u << j;
                if (inside && c <= 0.0f) holes += 1;
            }
            if (nzb) atomicOr(&s_rowbits[yl * 2 + (q >> 3)], nzb << ((q & 7) * 4));
            if (y < g.h) {
                const int so0 = (oy0 + it * 4 * PL_NW + wave * 4) * oh * 4, soc = (oy0 + it * 4 * PL_NW + wave * 4) * ch * 4;
                const int vo = (rw * oh + xq) * 4, voc = (rw * ch + xq) * 4;
                if (xq + 3 < g.w) {
                    const proj_v4i vx4 = {__float_as_int(vxa[0]), __float_as_int(vxa[1]), __float_as_int(vxa[2]), __float_as_int(vxa[3])};
                    const proj_v4i vy4 = {__float_as_int(vya[0]), __float_as_int(vya[1]), __float_as_int(vya[2]), __float_as_int(vya[3])};
                    const proj_v4i c4 = {__float_as_int(ca[0]), __float_as_int(ca[1]), __float_as_int(ca[2]), __float_as_int(ca[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(vx4, ro0, vo, so0, PROJ_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(vy4, ro1, vo, so0, PROJ_ST_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(c4, rcn, voc, soc, PROJ_ST_AUX);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (xq + j < g.w) {
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(vxa[j]), ro0, vo + 4 * j, so0, 0);
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(vya[j]), ro1, vo + 4 * j, so0, 0);
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ca[j]), rcn, voc + 4 * j, soc, 0);
                        }
                }
            }
        }
        if (holes) atomicAdd(&s_misc[1], holes);
        if (negs) atomicAdd(&s_misc[2], negs);
#ifdef PROJ_STAMPS
        if (!attempt) st_t2 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
        const int busiest = s_misc[0];
        if (busiest <= PROJ_ADD_CELL || attempt) break;
        // a cell took more addends than fit beside it: everything once more, with addends small enough for the busiest cell
        // (this thread's stores are overwritten in program order)
        shift = (32 - __clz(busiest - 1)) - 5;                  // ceil(log2(busiest)) - log2(32)
        __syncthreads();                                        // every thread has read the counter
        for (int i = tid; i < LY::total / 16; i += PL_THREADS) lds[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    pull_bitmaps<PL_THREADS>(g, ws, bits, s_rowbits, s_misc, b, txi, tyi, ox0, oy0, tile, tid);
#ifdef PROJ_STAMPS
    if (tid == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(ws + g.off_list + g.ntiles + (g.ntiles & 1)) + (int64_t)tile * 8;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        st[0] = st_t0; st[1] = st_t1; st[2] = st_t2; st[3] = __builtin_amdgcn_s_memtime();
        st[4] = st_r0; st[5] = __builtin_amdgcn_s_memrealtime(); st[6] = hwid; st[7] = (unsigned long long)xcc | ((st_ta - st_t0) << 8);
    }
#endif
}

// K2: pass 3 (flowprojection_cuda_kernel.cu:175-231).  A cell read here is either a non-hole
// (never written by this pass) or is multiplied by 0.
//
// The reference walks cell by cell from every hole until it meets a non-zero count; along an
// uncovered border strip that is a dependent chain of up to H (or W) loads per hole.  K1 has left
// row-packed and column-packed bitmaps of "count != 0" and a list of the tiles that have holes.  A
// workgroup takes a listed tile, one thread per cell.  Along the row: the wave of a row loads the row's
// whole bitmap line at once (lane i = word i) and every lane finds its neighbours from registers (a
// ballot says which words are non-zero, a lane exchange fetches the word).  Along the column: a lane
// loads its own word and the two beside it, and the whole line once to learn which words are non-zero
// (a neighbour further than a word away, rare, costs one more load).  Two dependent rounds of loads per
// hole -- lines, then the four cells found -- whatever the distance.  The cell found, hence the result,
// is the reference's.
struct ProjScan { int pos; float cnt; };

// cell-by-cell walk of the reference (fallback path: no bitmaps)
__device__ __forceinline__ ProjScan proj_walk_plain(const float* __restrict__ cn, int64_t origin, int64_t stride,
                                                    int p0, int len, int dir) {
    ProjScan r{p0, 0.0f};
    while (r.cnt == 0.0f && r.pos + dir >= 0 && r.pos + dir <= len - 1) { r.pos += dir; r.cnt = cn[origin + (int64_t)r.pos * stride]; }
    return r;
}

#define PROJ_FIN_THREADS 256        // a wave = four rows of the tile
#define PROJ_FIN_ROWS (PROJ_TH / (PROJ_FIN_THREADS / 64))
#define PROJ_LINE_CHUNKS 2          // row bitmap lines of up to 128 words (4096 pixels) are searched in registers

// nearest non-zero word strictly below / above word j of a line whose non-zero words are flagged in m[]; -1: none
__device__ __forceinline__ int mask_prev(const unsigned long long* m, int j) {
    int found = -1;
#pragma unroll
    for (int c = 0; c < PROJ_LINE_CHUNKS; ++c) {
        const int rel = j - 64 * c;                                 // words of this chunk below j: bits < rel
        const unsigned long long q = rel <= 0 ? 0ull : rel >= 64 ? m[c] : (m[c] & ((1ull << rel) - 1ull));
        if (q) found = 64 * c + 63 - __clzll((long long)q);
    }
    return found;
}
__device__ __forceinline__ int mask_next(const unsigned long long* m, int j) {
    int found = -1;
#pragma unroll
    for (int c = PROJ_LINE_CHUNKS - 1; c >= 0; --c) {
        const int rel = j - 64 * c;                                 // words of this chunk above j: bits > rel
        const unsigned long long q = rel >= 63 ? 0ull : rel < 0 ? m[c] : (m[c] & ~((2ull << rel) - 1ull));
        if (q) found = 64 * c + __ffsll((long long)q) - 1;
    }
    return found;
}

// Round 3: a tile is a 256-thread workgroup (lane = column, a wave = four rows) whose state words come through the
// scalar cache, so a tile without holes costs one scalar load; a tile with holes makes TWO rounds of vector loads --
// the bitmap words of its four rows and three words of its column (own, previous, next: a column's nearest covered
// cell further than 32 rows away, rare, walks the line), then the counts and flows of the cells found -- both rounds
// issued for the wave's four rows together.  (Round 2: 1024 threads, a wave per row, the whole column line per lane:
// 11.4 us per call at 1080p, of which ~5 are the launch and the wait for K1's writes.)
__global__ __launch_bounds__(PROJ_FIN_THREADS) void proj_finish(
    ProjDst dst, int per, ProjGeom g, vfi_strides s1, vfi_strides sc,
    int* __restrict__ ws, const int* __restrict__ bits, const float* __restrict__ planes, int fillhole) {
    const int tile = band_item(blockIdx.x, gridDim.x);
    // header words [2], [3] (the fallback's dirty extent) and the tile's word from K1 (0 = no holes, 1 = holes, 3 = holes
    // and negative counts: non-zero, yet holes -- the counts are read then)
    long long dirty;
    int tflag;
    asm volatile("s_load_dwordx2 %0, %2, 0x8\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(dirty), "=&s"(tflag) : "s"(ws), "s"(ws + g.off_list + tile) : "memory");
    const bool fallback = dirty != 0;
    if (tile == 0 && threadIdx.x == 0) ws[PROJ_WS_FLAG] = 0;               // K0 of the next call starts afresh
    // (and the tile's word: every word of this buffer is zero between calls, so a call on another frame size, whose records
    //  lie where this call's list was, finds them empty)
    if (threadIdx.x == 0 && tflag != 0) ws[g.off_list + tile] = 0;
    if (!fallback && (!fillhole || tflag == 0)) return;
    const int per_img = g.tiles_x * g.tiles_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool generic = g.rmw > 64 * PROJ_LINE_CHUNKS;
    const bool negatives = !fallback && (tflag & 2) != 0;
    const int b = tile / per_img;
    const int trem = tile - b * per_img;
    const int tyi = trem / g.tiles_x, txi = trem - tyi * g.tiles_x;
    const int x = txi * PROJ_TW + lane, y0 = tyi * PROJ_TH + wave * PROJ_FIN_ROWS;
    const ProjImage im = proj_image(b, per);
    float* const count = dst.count[im.item] + (int64_t)im.bi * sc.b;
    float* o0 = dst.out[im.item] + (int64_t)im.bi * s1.b;
    float* o1 = o0 + s1.c;
    if (fallback) {
        // K1 left sums in the scratch planes: normalise (pass 2) and fill holes (pass 3) from them
        if (x >= g.w) return;
        const int64_t npx = (int64_t)(g.ntiles / per_img) * g.h * g.w;
        const float* p0 = planes + (int64_t)b * g.h * g.w;
        const float* p1 = p0 + npx;
        const float* pc = p1 + npx;
        for (int k = 0; k < PROJ_FIN_ROWS; ++k) {
            const int y = y0 + k;
            if (y >= g.h) break;
            const int64_t row = (int64_t)y * s1.h;
            const int64_t me = (int64_t)y * g.w + x;
            const float c = pc[me];
            count[(int64_t)y * sc.h + x] = c;
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
        }
        return;
    }
    const float* cn = count;
    const int xc = min(x, g.w - 1);
    const int* cl = bits + g.colmap + (b * g.w + xc) * g.cmw;
    int xl[PROJ_FIN_ROWS], xr[PROJ_FIN_ROWS], yu[PROJ_FIN_ROWS], yd[PROJ_FIN_ROWS];
    bool hole[PROJ_FIN_ROWS];
    if (generic) {
#pragma unroll
        for (int k = 0; k < PROJ_FIN_ROWS; ++k) {
            const int y = y0 + k;
            hole[k] = x < g.w && y < g.h && cn[(int64_t)y * sc.h + x] <= 0.0f;
            xl[k] = xr[k] = yu[k] = yd[k] = -1;
            if (hole[k]) {
                const int* rl = bits + g.rowmap + (b * g.h + y) * g.rmw;
                xl[k] = proj_bit_walk(rl, x, g.w, -1); xr[k] = proj_bit_walk(rl, x, g.w, +1);
                yu[k] = proj_bit_walk(cl, y, g.h, -1); yd[k] = proj_bit_walk(cl, y, g.h, +1);
            }
        }
    } else {
        // ---- round one: the rows' lines (lane i = word 64 c + i), this lane's column words, and with negative counts its cells
        const int i = y0 >> 5;                                              // the column word of the tile's rows
        unsigned rwd[PROJ_FIN_ROWS][PROJ_LINE_CHUNKS];
        float cme[PROJ_FIN_ROWS];
#pragma unroll
        for (int k = 0; k < PROJ_FIN_ROWS; ++k) {
            const int* rl = bits + g.rowmap + (b * g.h + min(y0 + k, g.h - 1)) * g.rmw;
#pragma unroll
            for (int c = 0; c < PROJ_LINE_CHUNKS; ++c) rwd[k][c] = (64 * c + lane < g.rmw) ? (unsigned)rl[64 * c + lane] : 0u;
            cme[k] = (negatives && x < g.w && y0 + k < g.h) ? cn[(int64_t)(y0 + k) * sc.h + x] : 0.0f;
        }
        const int cwords = (g.h + 31) >> 5;
        const unsigned cown = (unsigned)cl[i], cprev = i > 0 ? (unsigned)cl[i - 1] : 0u, cnext = i + 1 < cwords ? (unsigned)cl[i + 1] : 0u;
        const int j = x >> 5, jb = x & 31;
#pragma unroll
        for (int k = 0; k < PROJ_FIN_ROWS; ++k) {
            const int y = y0 + k;
            unsigned long long rm[PROJ_LINE_CHUNKS];
#pragma unroll
            for (int c = 0; c < PROJ_LINE_CHUNKS; ++c) rm[c] = __ballot(rwd[k][c] != 0u);
            const int jl = mask_prev(rm, j), jr = mask_next(rm, j);
            // the words themselves come from the lanes that hold them
            unsigned own = 0u, wl = 0u, wr = 0u;
#pragma unroll
            for (int c = 0; c < PROJ_LINE_CHUNKS; ++c) {
                const unsigned a = (unsigned)__shfl((int)rwd[k][c], j & 63), l2 = (unsigned)__shfl((int)rwd[k][c], jl & 63), r2 = (unsigned)__shfl((int)rwd[k][c], jr & 63);
                if ((j >> 6) == c) own = a;
                if (jl >= 0 && (jl >> 6) == c) wl = l2;
                if (jr >= 0 && (jr >> 6) == c) wr = r2;
            }
            // a hole: its bit is clear -- or, in a tile that holds negative counts, its count is not positive
            hole[k] = x < g.w && y < g.h && (negatives ? cme[k] <= 0.0f : ((own >> jb) & 1u) == 0u);
            const unsigned below = own & ((1u << jb) - 1u), above = jb == 31 ? 0u : own & ~((2u << jb) - 1u);
            xl[k] = below ? j * 32 + 31 - __clz(below) : jl >= 0 ? jl * 32 + 31 - __clz(wl) : -1;
            xr[k] = above ? j * 32 + __ffs((int)above) - 1 : jr >= 0 ? jr * 32 + __ffs((int)wr) - 1 : -1;
            const int ib = y & 31;
            const unsigned cbelow = cown & ((1u << ib) - 1u), cabove = ib == 31 ? 0u : cown & ~((2u << ib) - 1u);
            yu[k] = cbelow ? i * 32 + 31 - __clz(cbelow) : cprev ? (i - 1) * 32 + 31 - __clz(cprev) : -2;
            yd[k] = cabove ? i * 32 + __ffs((int)cabove) - 1 : cnext ? (i + 1) * 32 + __ffs((int)cnext) - 1 : -2;
            // not within the three words: walk the rest of the line (32 and more uncovered rows on end)
            if (hole[k] && yu[k] == -2) yu[k] = i > 1 ? proj_bit_walk(cl, (i - 1) * 32, g.h, -1) : -1;
            if (hole[k] && yd[k] == -2) yd[k] = i + 2 < cwords ? proj_bit_walk(cl, (i + 1) * 32 + 31, g.h, +1) : -1;
        }
    }
    // ---- round two: the cells found.  A walk that found nothing contributes weight 0; its position only has to be valid
    float lc[PROJ_FIN_ROWS], rc[PROJ_FIN_ROWS], uc[PROJ_FIN_ROWS], dc[PROJ_FIN_ROWS];
    float a0[PROJ_FIN_ROWS], b0[PROJ_FIN_ROWS], c0[PROJ_FIN_ROWS], d0[PROJ_FIN_ROWS], a1[PROJ_FIN_ROWS], b1[PROJ_FIN_ROWS], c1[PROJ_FIN_ROWS], d1[PROJ_FIN_ROWS];
#pragma unroll
    for (int k = 0; k < PROJ_FIN_ROWS; ++k) {
        lc[k] = rc[k] = uc[k] = dc[k] = 0.0f;
        a0[k] = b0[k] = c0[k] = d0[k] = a1[k] = b1[k] = c1[k] = d1[k] = 0.0f;
        if (hole[k]) {
            const int y = y0 + k;
            const int64_t row = (int64_t)y * s1.h, crow = (int64_t)y * sc.h;
            const int pl_ = xl[k] < 0 ? x : xl[k], pr_ = xr[k] < 0 ? x : xr[k], pu = yu[k] < 0 ? y : yu[k], pd = yd[k] < 0 ? y : yd[k];
            if (xl[k] >= 0) lc[k] = cn[crow + xl[k]];
            if (xr[k] >= 0) rc[k] = cn[crow + xr[k]];
            if (yu[k] >= 0) uc[k] = cn[(int64_t)yu[k] * sc.h + x];
            if (yd[k] >= 0) dc[k] = cn[(int64_t)yd[k] * sc.h + x];
            a0[k] = o0[row + pl_]; b0[k] = o0[row + pr_]; c0[k] = o0[(int64_t)pu * s1.h + x]; d0[k] = o0[(int64_t)pd * s1.h + x];
            a1[k] = o1[row + pl_]; b1[k] = o1[row + pr_]; c1[k] = o1[(int64_t)pu * s1.h + x]; d1[k] = o1[(int64_t)pd * s1.h + x];
        }
    }
#pragma unroll
    for (int k = 0; k < PROJ_FIN_ROWS; ++k) {
        if (!hole[k] || lc[k] + rc[k] + uc[k] + dc[k] <= 0.0f) continue;
        const float lt = (lc[k] > 0.0f) ? 1.0f : 0.0f;
        const float rt = (rc[k] > 0.0f) ? 1.0f : 0.0f;
        const float ut = (uc[k] > 0.0f) ? 1.0f : 0.0f;
        const float dt = (dc[k] > 0.0f) ? 1.0f : 0.0f;
        const float den = lt + rt + ut + dt;
        const int64_t row = (int64_t)(y0 + k) * s1.h;
        o0[row + x] = (lt * a0[k] + rt * b0[k] + ut * c0[k] + dt * d0[k]) / den;
        o1[row + x] = (lt * a1[k] + rt * b1[k] + ut * c1[k] + dt * d1[k]) / den;
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

// Geometry and workspace sizes of a call; false when the frame is beyond what the records can index.
struct ProjSizes { size_t words, bit_words, plane_floats; };
static bool proj_geometry(int batch, int h, int w, ProjGeom* g, ProjSizes* z) {
    if (batch <= 0 || h <= 0 || w <= 0 || h > 32767 || w > 32767) return false;
    g->h = h; g->w = w;
    g->tiles_x = (w + PROJ_TW - 1) / PROJ_TW;
    g->tiles_y = (h + PROJ_TH - 1) / PROJ_TH;
    const int64_t nt = (int64_t)g->tiles_x * g->tiles_y * batch;
    if (nt > (1 << 24)) return false;
    g->ntiles = (int)nt;
    g->rmw = (w + 31) / 32;
    g->cmw = (((g->tiles_y * PROJ_TH + 31) / 32) + 3) & ~3;     // whole tiles (K1 stores 16-bit halves), whole 16-byte groups
    const size_t row_words = ((size_t)batch * h * g->rmw + 3) & ~(size_t)3;
    z->bit_words = row_words + (size_t)batch * w * g->cmw;
    if (z->bit_words > (size_t)INT_MAX) return false;
    g->rowmap = 0;
    g->colmap = (int)row_words;
    g->off_tile = PROJ_WS_HDR;
    g->off_list = g->off_tile + PROJ_TILE_WORDS * g->ntiles;
    z->words = (size_t)g->off_list + (size_t)nt;
#ifdef PROJ_STAMPS
    z->words += 2 + 16 * (size_t)nt;
#endif
    z->plane_floats = (size_t)3 * batch * h * w;
    return true;
}

struct ProjBuffers { int* words; int* bits; float* planes; };
static bool proj_buffers(hipStream_t st, const ProjSizes& z, ProjBuffers* p) {
    // The header and the tile records (zero between calls) say what the scratch planes hold (zero between calls unless the
    // header says otherwise), and the bitmaps go with them: the three buffers are one set, allocated and retired together
    // (workspace.h) -- with separate lifetimes a graph captured before the planes grew would keep the old planes beside a
    // header that later calls clear after cleaning the NEW planes.
    const WsSlot slots[3] = {WS_PROJ_WORDS, WS_PROJ_BITS, WS_PROJ_PLANES};
    const size_t bytes[3] = {z.words * sizeof(int), z.bit_words * sizeof(int), z.plane_floats * sizeof(float)};
    const bool zero[3] = {true, false, true};
    void* ptrs[3];
    if (!ws_get_group(st, slots, bytes, zero, 3, ptrs)) return false;
    p->words = static_cast<int*>(ptrs[0]);
    p->bits = static_cast<int*>(ptrs[1]);
    p->planes = static_cast<float*>(ptrs[2]);
    return true;
}

// every in-plane element offset of a [*, *, h, w] tensor with row stride sh fits 31 bits
static bool fits32(int64_t sh, int h, int w) { return sh >= 0 && sh * (int64_t)(h - 1) + w < ((int64_t)1 << 31); }

// The launch triple for n <= PROJ_NMAX items of `per` images each.  sf = strides of the flows, s1 = strides of the outputs
// (the reference binding shares them), s2 / sc = strides of the depths / counts.
template <bool DEPTH>
static int project_forward_list(const float* const* flows, vfi_strides sf, const float* const* depths, float* const* counts,
                                float* const* outs, int n, int per, int h, int w, int fillhole, vfi_strides s1, vfi_strides s2,
                                vfi_strides sc, hipStream_t st) {
    if (n <= 0 || n > PROJ_NMAX || per <= 0 || (int64_t)n * per > INT_MAX) return VFI_ERR_SHAPE;
    const int images = n * per;
    ProjGeom g;
    ProjSizes z;
    if (!proj_geometry(images, h, w, &g, &z)) return VFI_ERR_SHAPE;
    if (!fits32(sf.h, h, w) || !fits32(s1.h, h, w) || !fits32(sc.h, h, w) || (DEPTH && !fits32(s2.h, h, w)))
        return VFI_ERR_SHAPE;
    ProjBuffers p;
    if (!proj_buffers(st, z, &p)) return VFI_ERR_LAUNCH;
    ProjSrc src;
    ProjDst dst;
    // 16-byte lanes need 16-byte aligned rows
    bool vec_in = sf.b % 4 == 0 && sf.c % 4 == 0 && sf.h % 4 == 0 && (!DEPTH || (s2.b % 4 == 0 && s2.h % 4 == 0));
    bool vec_out = s1.b % 4 == 0 && s1.c % 4 == 0 && s1.h % 4 == 0 && sc.b % 4 == 0 && sc.h % 4 == 0;
    for (int i = 0; i < PROJ_NMAX; ++i) {
        const int k = i < n ? i : 0;                        // (unused slots repeat item 0)
        if (!flows[k] || !counts[k] || !outs[k] || (DEPTH && !depths[k])) return VFI_ERR_SHAPE;
        src.flow[i] = flows[k]; src.depth[i] = DEPTH ? depths[k] : nullptr;
        dst.count[i] = counts[k]; dst.out[i] = outs[k];
        vec_in = vec_in && (uintptr_t)flows[k] % 16 == 0 && (!DEPTH || (uintptr_t)depths[k] % 16 == 0);
        vec_out = vec_out && (uintptr_t)outs[k] % 16 == 0 && (uintptr_t)counts[k] % 16 == 0;
    }
    // every item's count and output are written in the same launch: they must not share memory
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (counts[i] == counts[j] || outs[i] == outs[j]) return VFI_ERR_SHAPE;
    src.fb = sf.b; src.fc = sf.c; src.db = DEPTH ? s2.b : 0;
    src.fh = (int)sf.h; src.dh = DEPTH ? (int)s2.h : 0;
    src.per = per;
    if (vec_in) {
        const int groups_x = (g.tiles_x + 3) / 4;
        hipLaunchKernelGGL((proj_scan4<DEPTH>), dim3(images * g.tiles_y * groups_x), dim3(256), 0, st, src, g, groups_x,
                           p.words, p.planes, (int64_t)z.plane_floats);
    } else {
        hipLaunchKernelGGL((proj_scan<DEPTH>), dim3(g.ntiles), dim3(64), 0, st, src, g, p.words, p.planes,
                           (int64_t)z.plane_floats);
    }
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    // 16-byte stores need 16-byte aligned rows
    const int vec_ok = vec_out ? 1 : 0;
#ifndef PROJ_NO_LEAN
    if (vec_in && vec_ok && g.tiles_y <= 65535 && images <= 65535)
        hipLaunchKernelGGL((proj_pull_lean<DEPTH>), PROJ_BANDS ? dim3(g.ntiles) : dim3(g.tiles_x, g.tiles_y, images), dim3(PL_THREADS), 0, st, src, dst, g,
                           (int64_t)s1.b, (int64_t)s1.c, (int)s1.h, (int64_t)sc.b, (int)sc.h, p.words, p.bits, p.planes, (int64_t)z.plane_floats);
    else
#endif
    if (vec_in)
        hipLaunchKernelGGL((proj_pull<DEPTH, true>), dim3(g.ntiles), dim3(PROJ_PULL_THREADS), 0, st, src, dst, g,
                           (int64_t)s1.b, (int64_t)s1.c, (int)s1.h, (int64_t)sc.b, (int)sc.h, vec_ok, p.words, p.bits, p.planes, (int64_t)z.plane_floats);
    else
        hipLaunchKernelGGL((proj_pull<DEPTH, false>), dim3(g.ntiles), dim3(PROJ_PULL_THREADS), 0, st, src, dst, g,
                           (int64_t)s1.b, (int64_t)s1.c, (int)s1.h, (int64_t)sc.b, (int)sc.h, vec_ok, p.words, p.bits, p.planes, (int64_t)z.plane_floats);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    // (also runs with fillhole == 0: it resets the call's state, and the fallback path normalises there)
    hipLaunchKernelGGL(proj_finish, dim3(g.ntiles), dim3(PROJ_FIN_THREADS), 0, st, dst, per, g,
                       s1, sc, p.words, p.bits, p.planes, fillhole);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    return VFI_OK;
}

// any number of items: groups of PROJ_NMAX, one launch triple each
template <bool DEPTH>
static int project_forward_items(const float* const* flows, vfi_strides sf, const float* const* depths, float* const* counts,
                                 float* const* outs, int n, int per, int h, int w, int fillhole, vfi_strides s1, vfi_strides s2,
                                 vfi_strides sc, hipStream_t st) {
    if (n <= 0 || !flows || !counts || !outs || (DEPTH && !depths)) return VFI_ERR_SHAPE;
    for (int i0 = 0; i0 < n; i0 += PROJ_NMAX) {
        const int m = n - i0 < PROJ_NMAX ? n - i0 : PROJ_NMAX;
        const int err = project_forward_list<DEPTH>(flows + i0, sf, DEPTH ? depths + i0 : nullptr, counts + i0, outs + i0, m, per, h, w,
                                                    fillhole, s1, s2, sc, st);
        if (err != VFI_OK) return err;
    }
    return VFI_OK;
}

// the reference bindings' call: one flow tensor
template <bool DEPTH>
static int project_forward(const float* flow, vfi_strides sf, const float* in2, float* count, float* out, int batch, int h, int w,
                           int fillhole, vfi_strides s1, vfi_strides s2, vfi_strides sc, hipStream_t st) {
    return project_forward_list<DEPTH>(&flow, sf, &in2, &count, &out, 1, batch, h, w, fillhole, s1, s2, sc, st);
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

#ifdef PROJ_STAMPS
extern "C" int vfi_dev_projection_stamps(int batch, int h, int w, vfi_stream_t stream, unsigned long long* host_out) {
    ProjGeom g;
    ProjSizes z;
    if (!proj_geometry(batch, h, w, &g, &z)) return VFI_ERR_SHAPE;
    ProjBuffers p;
    if (!proj_buffers((hipStream_t)stream, z, &p)) return VFI_ERR_LAUNCH;
    (void)hipDeviceSynchronize();
    return hipMemcpy(host_out, p.words + g.off_list + g.ntiles + (g.ntiles & 1), (size_t)g.ntiles * 64, hipMemcpyDeviceToHost) == hipSuccess
               ? VFI_OK : VFI_ERR_LAUNCH;
}
#endif

extern "C" int vfi_projection_reserve(int batch, int h, int w, vfi_stream_t stream) {
    ProjGeom g;
    ProjSizes z;
    if (!proj_geometry(batch, h, w, &g, &z)) return VFI_ERR_SHAPE;
    ProjBuffers p;
    if (!proj_buffers((hipStream_t)stream, z, &p)) return VFI_ERR_LAUNCH;
    // (and the scratch tensor of the *_forward_up4 entry points, should the caller capture one of those)
    return ws_get((hipStream_t)stream, WS_PROJ_UPFLOW, (size_t)batch * 2 * h * w * sizeof(float), false, nullptr) ? VFI_OK : VFI_ERR_LAUNCH;
}

extern "C" int vfi_flowprojection_forward(const float* input1, float* count, float* output, int batch, int h, int w,
                                           int fillhole, vfi_strides s1, vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !count || !output) return VFI_ERR_SHAPE;
    return project_forward<false>(input1, s1, nullptr, count, output, batch, h, w, fillhole, s1, s1, sc, (hipStream_t)stream);
}

extern "C" int vfi_depthflowprojection_forward(const float* input1, const float* input2, float* count, float* output,
                                                int batch, int h, int w, int fillhole, vfi_strides s1, vfi_strides s2,
                                                vfi_strides sc, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !count || !output) return VFI_ERR_SHAPE;
    return project_forward<true>(input1, s1, input2, count, output, batch, h, w, fillhole, s1, s2, sc, (hipStream_t)stream);
}

// FlowProject(inputs, depth) of the networks: every flow of the list (and, when the caller concatenates them, of both directions)
// in one launch triple per PROJ_NMAX items
extern "C" int vfi_flowprojection_forward_batch(const float* const* inputs1, float* const* counts, float* const* outputs, int nitems,
                                                 int batch, int h, int w, int fillhole, vfi_strides s1, vfi_strides sc,
                                                 vfi_stream_t stream) {
    if (nitems <= 0 || batch <= 0 || h <= 0 || w <= 0 || !inputs1 || !counts || !outputs) return VFI_ERR_SHAPE;
    return project_forward_items<false>(inputs1, s1, nullptr, counts, outputs, nitems, batch, h, w, fillhole, s1, s1, sc, (hipStream_t)stream);
}

extern "C" int vfi_depthflowprojection_forward_batch(const float* const* inputs1, const float* const* inputs2, float* const* counts,
                                                      float* const* outputs, int nitems, int batch, int h, int w, int fillhole,
                                                      vfi_strides s1, vfi_strides s2, vfi_strides sc, vfi_stream_t stream) {
    if (nitems <= 0 || batch <= 0 || h <= 0 || w <= 0 || !inputs1 || !inputs2 || !counts || !outputs) return VFI_ERR_SHAPE;
    return project_forward_items<true>(inputs1, s1, inputs2, counts, outputs, nitems, batch, h, w, fillhole, s1, s2, sc, (hipStream_t)stream);
}

// ---- fused glue: forward_flownets + FlowProject (networks/DAIN_slowmotion.py:204-216, 301-308)
extern "C" int vfi_flow_upsample4(const float* input, float* output, int batch, int channels, int hq, int wq,
                                   float mul0, float mul1, vfi_strides sq, vfi_strides so, vfi_stream_t stream) {
    if (batch <= 0 || channels <= 0 || hq <= 0 || wq <= 0 || hq > INT_MAX / 4 || wq > INT_MAX / 4 || !input || !output)
        return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(flow_upsample4, pixel_grid(4 * wq, 4 * hq, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       input, output, channels, hq, wq, mul0, mul1, sq, so);
    return launch_status();
}

// The quarter-resolution flow of the network -> its projection, in one call: x4 upsample of (mul0 * flow) * mul1 into a
// per-stream scratch tensor, then the projection above.  (Rounds 1-2 formed the upsample inside the projection kernels'
// source reads -- the full-resolution flow never existed -- at 4-byte lanes and eight taps per source pixel: 108 us at 1080p.
// With round 3's kernels the two steps take 10 + 31 us and the 18 MB round trip through the caches costs less than
// the fusion saved; same arithmetic, so the same bits as vfi_flow_upsample4 followed by vfi_[depth]flowprojection_forward.)
template <bool DEPTH>
static int project_forward_up4(const float* flow_q, vfi_strides sq, int hq, int wq, float m0, float m1, const float* in2,
                               float* count, float* out, int batch, int fillhole, vfi_strides so, vfi_strides s2, vfi_strides sc,
                               hipStream_t st) {
    const int h = 4 * hq, w = 4 * wq;
    const int64_t plane = (int64_t)h * w;
    float* full = static_cast<float*>(ws_get(st, WS_PROJ_UPFLOW, (size_t)batch * 2 * plane * sizeof(float), false, nullptr));
    if (!full) return VFI_ERR_LAUNCH;
    const vfi_strides sfull{2 * plane, plane, (int64_t)w};
    hipLaunchKernelGGL(flow_upsample4, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, st, flow_q, full, 2, hq, wq, m0, m1, sq, sfull);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    return project_forward<DEPTH>(full, sfull, in2, count, out, batch, h, w, fillhole, so, s2, sc, st);
}

extern "C" int vfi_flowprojection_forward_up4(const float* flow_q, float* count, float* output, int batch, int hq, int wq,
                                               float mul0, float mul1, int fillhole, vfi_strides sq, vfi_strides sc,
                                               vfi_strides so, vfi_stream_t stream) {
    if (batch <= 0 || hq <= 0 || wq <= 0 || hq > INT_MAX / 4 || wq > INT_MAX / 4 || !flow_q || !count || !output)
        return VFI_ERR_SHAPE;
    return project_forward_up4<false>(flow_q, sq, hq, wq, mul0, mul1, nullptr, count, output, batch, fillhole, so, so, sc,
                                      (hipStream_t)stream);
}

extern "C" int vfi_depthflowprojection_forward_up4(const float* flow_q, const float* input2, float* count, float* output,
                                                    int batch, int hq, int wq, float mul0, float mul1, int fillhole,
                                                    vfi_strides sq, vfi_strides s2, vfi_strides sc, vfi_strides so,
                                                    vfi_stream_t stream) {
    if (batch <= 0 || hq <= 0 || wq <= 0 || hq > INT_MAX / 4 || wq > INT_MAX / 4 || !flow_q || !input2 || !count || !output)
        return VFI_ERR_SHAPE;
    return project_forward_up4<true>(flow_q, sq, hq, wq, mul0, mul1, input2, count, output, batch, fillhole, so, s2, sc,
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
