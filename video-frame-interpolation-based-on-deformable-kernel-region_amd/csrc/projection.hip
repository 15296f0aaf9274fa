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
// How (owner computes; no global atomics on the normal path): the frame is cut
// into 64x16 tiles, used both as source tiles and as output tiles.
//   A  proj_bin     one wave per SOURCE row segment (64 pixels of one row): bounding
//                   box of the segment's targets -> the segment appends its id to
//                   the list of every OUTPUT tile that box touches (~2 MB of lists
//                   in a workspace; lists hold PROJ_LIST_CAP entries).
//   B  proj_gather  one workgroup per OUTPUT tile: accumulates the three planes of
//                   its tile in LDS from the row segments on its list (one wave per
//                   segment), then writes count and the normalised flow once,
//                   coalesced.
//   D  proj_finish  hole filling (and, on the fallback path, normalisation)
// Launch order A, B, D.  B (or D on the fallback path) writes every cell of count and output,
// so callers need not zero-fill them (the reference's callers must: its splat accumulates
// into them).
// Fallback: when any list overflows (fields with displacements of many tiles,
// e.g. random flow of +-W/2), B instead splats its own source tile with global
// atomics exactly like the reference -- into three scratch planes of the workspace that
// are zero between calls -- and D normalises them into count / output and fills holes from
// them.  The switch is a serial number in the workspace written by A and read by B and D:
// no host round trip, and no launch that exists only for the fallback.  (The scratch planes
// are cleaned by the next call's A, whose workgroups see a "dirty" word.)
//
// Accumulation in B is 64-bit fixed point (ds_add_u64): on gfx950 an LDS float atomic add
// costs ~170 cycles per wave instruction (measured, tools/probes/lds_atomic_probe.hip), an
// integer one ~6.  Every addend is scaled by a power of two chosen per output tile from the
// largest |addend| that reaches it (found by A), rounded to a 32-bit integer (an error below
// 2^-31 of that largest addend) and summed exactly in int64; the sum is converted back to
// float once.  The result does not depend on the summation order, so it is reproducible bit
// for bit from run to run (the reference's fp32 atomic sum carries one rounding per addend, in
// arrival order), and it agrees with any fp32 summation order to rounding; addends that are
// multiples of 2^-k (k < ~20) sum exactly in both.  count of FlowProjection is exact.
#include "vfi_common.h"
#include "bitwalk.h"

#include <limits.h>

#include <mutex>
#include <deque>

namespace vfi {

#define PROJ_TW 64
#define PROJ_TH 16
#define PROJ_THREADS 256
#define PROJ_LIST_CAP 252           // source row segments per output tile before the fallback kicks in
#define PROJ_SEG_SHIFT 12           // segment id = (batch * h + row) << 12 | tile column

// workspace "words" (32-bit): [0] serial of the last call whose lists overflowed; from word 16 one
// record per output tile: [0] list length, [1] / [2] bit patterns of the largest |value addend| /
// count addend among the listed segments (atomicMax by A), [3] unused, then PROJ_LIST_CAP segment
// ids.  B resets words 0..2 after reading them, so a record is empty between calls.
// workspace "bits": two bitmaps of "count != 0", one packed along rows (rowmap[b][y][x/32]) and
// one packed along columns (colmap[b][x][y/32]), written by B for the hole filler (A clears the
// column-packed one, whose words are shared by two tiles); its own allocation, so its layout
// (which depends on the frame size) cannot disturb the records.  A record's position depends on the tile index only, so calls with different frame
// sizes can share the workspace without stale lengths.
#define PROJ_WS_HDR 16
#define PROJ_WS_DIRTY 1             // header word: the scratch planes of the fallback hold sums
#define PROJ_WS_REC (4 + PROJ_LIST_CAP)
#define PROJ_REC_VMAX 1
#define PROJ_REC_CMAX 2
#define PROJ_REC_IDS 4
static inline size_t proj_ws_tile_words(int ntiles) { return PROJ_WS_HDR + (size_t)ntiles * PROJ_WS_REC; }

// rmw / cmw: 32-bit words per image row / column of the two "count != 0" bitmaps;
// rowmap / colmap: their word offsets inside the workspace's bit buffer
struct ProjGeom { int h, w, tiles_x, tiles_y, ntiles, rmw, cmw, rowmap, colmap; };

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

__device__ __forceinline__ int wmin(int v) { return wave_min_i32(v); }
__device__ __forceinline__ int wmax(int v) { return wave_max_i32(v); }

// A: bin source row segments into the lists of the output tiles they reach.  A workgroup covers the
// 64x16 tile of its index: its 16 row segments first find their target rectangles (in tiles), then
// one thread per candidate output tile reserves, with ONE returning atomic, room for all the
// segments that reach that tile and writes their ids (a returning atomic per segment and tile
// measured 4x slower: the round trips serialise).
#define PROJ_BIN_CAND 64            // candidate output tiles per source tile handled by the fast path
template <bool DEPTH, bool UP>
__global__ __launch_bounds__(PROJ_THREADS, 8) void proj_bin(
    ProjFlow flow, const float* __restrict__ in2, ProjGeom g, vfi_strides s2,
    int* __restrict__ ws, int* __restrict__ bits, float* __restrict__ planes, int64_t plane_floats, int serial) {
    if (ws[PROJ_WS_DIRTY] != 0) {
        // the previous call on this workspace took the fallback: its scratch planes are cleaned here,
        // a slice per workgroup (D clears the word at the end of this call)
        const int64_t chunk = (plane_floats + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * chunk, hi = min(plane_floats, lo + chunk);
        for (int64_t i = lo + threadIdx.x; i < hi; i += PROJ_THREADS) planes[i] = 0.0f;
    }
    __shared__ int rect[PROJ_TH][4];                        // per row segment: tx0, ty0, tx1, ty1 (tx0 < 0: none)
    __shared__ int tmax[2];
    const int tile = blockIdx.x;
    const int b = tile / (g.tiles_x * g.tiles_y);
    const int trem = tile - b * (g.tiles_x * g.tiles_y);
    const int tyi = trem / g.tiles_x, txi = trem - tyi * g.tiles_x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = txi * PROJ_TW + lane;
    // clear the column-packed bitmap words this tile shares with the tile below it (32 rows per word)
    if ((tyi & 1) == 0 && tid < PROJ_TW && x < g.w) bits[g.colmap + (b * g.w + x) * g.cmw + (tyi >> 1)] = 0;
    if (tid < 2) tmax[tid] = 0;
    __syncthreads();
    ProjSplat s[PROJ_TH / 4];
#pragma unroll
    for (int r = 0; r < PROJ_TH / 4; ++r)                   // all loads first
        s[r] = proj_source<DEPTH, UP>(flow, in2, b, x, tyi * PROJ_TH + wave + r * 4, g.h, g.w, s2);
    int vbits = 0, cbits = 0;
#pragma unroll
    for (int r = 0; r < PROJ_TH / 4; ++r) {
        const int x0 = wmin(s[r].valid ? s[r].L : INT_MAX), x1 = wmax(s[r].valid ? s[r].R : INT_MIN);
        const int y0 = wmin(s[r].valid ? s[r].T : INT_MAX), y1 = wmax(s[r].valid ? s[r].Bm : INT_MIN);
        // non-negative floats order like their bit patterns
        vbits = max(vbits, __float_as_int(s[r].valid ? fmaxf(fabsf(s[r].ax), fabsf(s[r].ay)) : 0.0f));
        cbits = max(cbits, __float_as_int(s[r].valid ? fabsf(s[r].ac) : 0.0f));
        if (lane == 0) {
            int* q = rect[wave + r * 4];
            const bool any = x0 != INT_MAX;
            q[0] = any ? x0 / PROJ_TW : -1; q[1] = any ? y0 / PROJ_TH : 0;
            q[2] = any ? x1 / PROJ_TW : -1; q[3] = any ? y1 / PROJ_TH : 0;
        }
    }
    vbits = wmax(vbits); cbits = wmax(cbits);
    if (lane == 0) { atomicMax(&tmax[0], vbits); atomicMax(&tmax[1], cbits); }
    __syncthreads();
    // union rectangle of the tile's segments
    int cx0 = INT_MAX, cy0 = INT_MAX, cx1 = INT_MIN, cy1 = INT_MIN;
#pragma unroll 2
    for (int r = 0; r < PROJ_TH; ++r)
        if (rect[r][0] >= 0) {
            cx0 = min(cx0, rect[r][0]); cy0 = min(cy0, rect[r][1]);
            cx1 = max(cx1, rect[r][2]); cy1 = max(cy1, rect[r][3]);
        }
    if (cx0 == INT_MAX) return;                             // nothing of this tile lands in the frame
    const int nx = cx1 - cx0 + 1, ncand = nx * (cy1 - cy0 + 1);
    const int vmax = tmax[0], cmax = tmax[1];
    // segment id = global row << 12 | tile column: the gather splits it with a shift and a mask
    const int seg0 = ((b * g.h + tyi * PROJ_TH) << PROJ_SEG_SHIFT) | txi;
    for (int c = tid; c < ncand; c += PROJ_THREADS) {
        const int ctx = cx0 + c % nx, cty = cy0 + c / nx;
        unsigned rows = 0u;                                 // which of the 16 segments reach this output tile
#pragma unroll 2
        for (int r = 0; r < PROJ_TH; ++r)
            if (rect[r][0] >= 0 && ctx >= rect[r][0] && ctx <= rect[r][2] && cty >= rect[r][1] && cty <= rect[r][3])
                rows |= 1u << r;
        if (!rows) continue;
        int* rec = ws + PROJ_WS_HDR + (int64_t)((b * g.tiles_y + cty) * g.tiles_x + ctx) * PROJ_WS_REC;
        int slot = atomicAdd(&rec[0], __popc(rows));
        if (slot + __popc(rows) > PROJ_LIST_CAP) { ws[0] = serial; continue; }    // overflow: fallback path
        while (rows) {
            const int r = __ffs((int)rows) - 1;
            rows &= rows - 1u;
            rec[PROJ_REC_IDS + slot++] = seg0 + (r << PROJ_SEG_SHIFT);
        }
        if (vmax > __hip_atomic_load(&rec[PROJ_REC_VMAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&rec[PROJ_REC_VMAX], vmax);
        if (cmax > __hip_atomic_load(&rec[PROJ_REC_CMAX], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&rec[PROJ_REC_CMAX], cmax);
    }
}

// B: one workgroup per output tile
template <bool DEPTH, bool UP>
__global__ __launch_bounds__(PROJ_THREADS) void proj_gather(
    ProjFlow flow, const float* __restrict__ in2, float* __restrict__ count, float* __restrict__ out,
    ProjGeom g, vfi_strides s1, vfi_strides s2, vfi_strides sc, int* __restrict__ ws, int* __restrict__ bits,
    float* __restrict__ planes, int serial) {
    __shared__ unsigned long long acc[3][PROJ_TH][PROJ_TW];
    const int tile = blockIdx.x;
    const int b = tile / (g.tiles_x * g.tiles_y);
    const int trem = tile - b * (g.tiles_x * g.tiles_y);
    const int tyi = trem / g.tiles_x, txi = trem - tyi * g.tiles_x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int* rec = ws + PROJ_WS_HDR + (int64_t)tile * PROJ_WS_REC;
    const int* list = rec + PROJ_REC_IDS;
    const int nsrc = min(rec[0], PROJ_LIST_CAP);
    const int vb = rec[PROJ_REC_VMAX], cb = rec[PROJ_REC_CMAX];
    const bool fallback = ws[0] == serial;
    // fixed-point scales: addend * 2^k rounded to a 32-bit integer, summed in int64.  Every addend that
    // reaches this tile is below 2^e with e from its list's maxima, so k = 30 - e keeps the product
    // inside int32 (one v_rndne + v_cvt instead of an emulated float -> int64), and a cell can take
    // 2^32 addends before the int64 sum overflows.  (Per output tile, so no global reduction is needed;
    // a sum only ever mixes addends of one scale.)
    int ev = 0, ec = 0;
    (void)frexpf(__int_as_float(vb), &ev);
    (void)frexpf(__int_as_float(cb), &ec);
    // (clamped so that 2^k stays a normal float when every addend is tiny or huge)
    const int kv = max(-100, min(100, 30 - ev)), kc = max(-100, min(100, 30 - ec));
    const float sv = ldexpf(1.0f, kv), scn = ldexpf(1.0f, kc);     // exact powers of two
    __syncthreads();                                        // every thread has read the record ...
    if (tid < 3) rec[tid] = 0;                              // ... leave it empty for the next call

    if (fallback) {
        // the reference's own scheme: this tile as SOURCE tile, global atomics into the dense scratch
        // planes [value x | value y | count][batch][h][w] of the workspace (zero between calls)
        const int64_t npx = (int64_t)(g.ntiles / (g.tiles_x * g.tiles_y)) * g.h * g.w;
        float* o0 = planes + (int64_t)b * g.h * g.w;
        float* o1 = o0 + npx;
        float* cn = o1 + npx;
#pragma unroll
        for (int r = 0; r < PROJ_TH / 4; ++r) {
            const ProjSplat s = proj_source<DEPTH, UP>(flow, in2, b, txi * PROJ_TW + lane, tyi * PROJ_TH + wave + r * 4,
                                                       g.h, g.w, s2);
            if (!s.valid) continue;
            const int64_t oT = (int64_t)s.T * g.w, oB = (int64_t)s.Bm * g.w;
            atomicAdd(&o0[oT + s.L], s.ax); atomicAdd(&o0[oT + s.R], s.ax); atomicAdd(&o0[oB + s.L], s.ax); atomicAdd(&o0[oB + s.R], s.ax);
            atomicAdd(&o1[oT + s.L], s.ay); atomicAdd(&o1[oT + s.R], s.ay); atomicAdd(&o1[oB + s.L], s.ay); atomicAdd(&o1[oB + s.R], s.ay);
            atomicAdd(&cn[oT + s.L], s.ac); atomicAdd(&cn[oT + s.R], s.ac); atomicAdd(&cn[oB + s.L], s.ac); atomicAdd(&cn[oB + s.R], s.ac);
        }
        return;
    }

    for (int i = tid; i < 3 * PROJ_TH * PROJ_TW; i += PROJ_THREADS) (&acc[0][0][0])[i] = 0ull;
    __syncthreads();
    const int ox0 = txi * PROJ_TW, oy0 = tyi * PROJ_TH;
    // One wave per source row segment.  A wave first fetches all its list entries with one load
    // (entry j of the wave sits in lane j), then walks them with the NEXT segment's pixels already in
    // flight: no dependent list -> pixel load chain per segment.
    const int mine = (nsrc > wave) ? (nsrc - wave + PROJ_THREADS / 64 - 1) / (PROJ_THREADS / 64) : 0;   // <= 63
    const int ids = (lane < mine) ? list[wave + lane * (PROJ_THREADS / 64)] : 0;
    auto fetch = [&](int j) {
        // wave-uniform: the row arithmetic of the loads stays on the scalar unit
        const int seg = __builtin_amdgcn_readlane(ids, j);
        return proj_load<DEPTH, UP>(flow, in2, b, (seg & ((1 << PROJ_SEG_SHIFT) - 1)) * PROJ_TW + lane,
                                    (seg >> PROJ_SEG_SHIFT) - b * g.h, g.h, g.w, s2);
    };
    ProjRaw nxt = fetch(0);                                 // lane 0 holds 0 when the wave has no entry: harmless
    for (int j = 0; j < mine; ++j) {
        const ProjRaw cur = nxt;
        if (j + 1 < mine) nxt = fetch(j + 1);               // in flight while this segment is accumulated
        const ProjSplat s = proj_make<DEPTH, UP>(flow, cur, g.h, g.w);
        // cells of this output tile only; R == L / Bm == T at the far edges add twice (:72-73)
        const int lx = s.L - ox0, rx = s.R - ox0, ty = s.T - oy0, by = s.Bm - oy0;
        const bool inL = s.valid && (unsigned)lx < PROJ_TW, inR = s.valid && (unsigned)rx < PROJ_TW;
        const bool inT = (unsigned)ty < PROJ_TH, inB = (unsigned)by < PROJ_TH;
        // addend * 2^k is exact in float (power-of-two scale) and below 2^30 in magnitude; two's
        // complement: adding the unsigned image of a negative int64 subtracts
        const unsigned long long qx = (unsigned long long)(long long)__float2int_rn(s.ax * sv);
        const unsigned long long qy = (unsigned long long)(long long)__float2int_rn(s.ay * sv);
        const unsigned long long qc = (unsigned long long)(long long)__float2int_rn(s.ac * scn);
        if (inT && inL) { atomicAdd(&acc[0][ty][lx], qx); atomicAdd(&acc[1][ty][lx], qy); atomicAdd(&acc[2][ty][lx], qc); }
        if (inT && inR) { atomicAdd(&acc[0][ty][rx], qx); atomicAdd(&acc[1][ty][rx], qy); atomicAdd(&acc[2][ty][rx], qc); }
        if (inB && inL) { atomicAdd(&acc[0][by][lx], qx); atomicAdd(&acc[1][by][lx], qy); atomicAdd(&acc[2][by][lx], qc); }
        if (inB && inR) { atomicAdd(&acc[0][by][rx], qx); atomicAdd(&acc[1][by][rx], qy); atomicAdd(&acc[2][by][rx], qc); }
    }
    __syncthreads();
    // normalise (flowprojection_cuda_kernel.cu:129-134) and write the tile once; leave the two
    // "count != 0" bitmaps for the hole filler
    __shared__ unsigned colm[PROJ_TW];
    if (tid < PROJ_TW) colm[tid] = 0u;
    __syncthreads();
    const int x = ox0 + lane;
    {
        unsigned mine = 0u;
#pragma unroll
        for (int r = 0; r < PROJ_TH / 4; ++r) {
            const int yl = wave + r * 4;
            const bool nz = acc[2][yl][lane] != 0ull && x < g.w && oy0 + yl < g.h;
            const unsigned long long rowbits = __ballot(nz);
            if (lane < 2 && oy0 + yl < g.h && txi * 2 + lane < g.rmw)
                bits[g.rowmap + (b * g.h + oy0 + yl) * g.rmw + txi * 2 + lane] = (int)(unsigned)(rowbits >> (32 * lane));
            if (nz) mine |= 1u << yl;
        }
        if (mine) atomicOr(&colm[lane], mine);
    }
    __syncthreads();
    if (tid < PROJ_TW && colm[tid] && ox0 + tid < g.w)
        atomicOr(&bits[g.colmap + (b * g.w + ox0 + tid) * g.cmw + (tyi >> 1)], (int)(colm[tid] << ((tyi & 1) * 16)));
    if (x < g.w) {
#pragma unroll
        for (int r = 0; r < PROJ_TH / 4; ++r) {
            const int yl = wave + r * 4, y = oy0 + yl;
            if (y >= g.h) continue;
            // exact integer sums -> float once (through double: one rounding)
            const float c = (float)ldexp((double)(long long)acc[2][yl][lane], -kc);
            float vx = (float)ldexp((double)(long long)acc[0][yl][lane], -kv);
            float vy = (float)ldexp((double)(long long)acc[1][yl][lane], -kv);
            if (c > 0.0f) { vx /= c; vy /= c; }
            float* o = out + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
            o[0] = vx;
            o[s1.c] = vy;
            count[(int64_t)b * sc.b + (int64_t)y * sc.h + x] = c;
        }
    }
}

// D: pass 3 (flowprojection_cuda_kernel.cu:175-231).  A cell read here is either a non-hole
// (never written by this pass) or is multiplied by 0.
//
// The reference walks cell by cell from every hole until it meets a non-zero count; along an
// uncovered border strip that is a dependent chain of up to H (or W) loads per hole.  On the
// normal path B has left row-packed and column-packed bitmaps of "count != 0", so a walk is a
// few word loads and a count-leading/trailing-zeros; only the cell found is then read.  The cell
// found -- hence the result -- is the same.
struct ProjScan { int pos; float cnt; };

// cell-by-cell walk of the reference (fallback path: no bitmaps)
__device__ __forceinline__ ProjScan proj_walk_plain(const float* __restrict__ cn, int64_t origin, int64_t stride,
                                                    int p0, int len, int dir) {
    ProjScan r{p0, 0.0f};
    while (r.cnt == 0.0f && r.pos + dir >= 0 && r.pos + dir <= len - 1) { r.pos += dir; r.cnt = cn[origin + (int64_t)r.pos * stride]; }
    return r;
}

__global__ __launch_bounds__(VFI_TX * VFI_TY) void proj_finish(
    float* __restrict__ count, float* out, ProjGeom g, vfi_strides s1, vfi_strides sc,
    int* __restrict__ ws, const int* __restrict__ bits, const float* __restrict__ planes, int fillhole, int serial) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    const int b = blockIdx.z;
    const bool fallback = ws[0] == serial;
    if (blockIdx.x == 0 && blockIdx.y == 0 && b == 0 && threadIdx.x == 0 && threadIdx.y == 0)
        ws[PROJ_WS_DIRTY] = fallback ? 1 : 0;               // read by the next call's A
    if (x >= g.w || y >= g.h) return;
    float* o0 = out + (int64_t)b * s1.b;
    float* o1 = o0 + s1.c;
    const int64_t row = (int64_t)y * s1.h;
    if (fallback) {
        // B left sums in the scratch planes: normalise (pass 2) and fill holes (pass 3) from them
        const int64_t npx = (int64_t)gridDim.z * g.h * g.w;
        const float* p0 = planes + (int64_t)b * g.h * g.w;
        const float* p1 = p0 + npx;
        const float* pc = p1 + npx;
        const int64_t me = (int64_t)y * g.w + x;
        const float c = pc[me];
        count[(int64_t)b * sc.b + (int64_t)y * sc.h + x] = c;
        if (c > 0.0f) {
            o0[row + x] = p0[me] / c;
            o1[row + x] = p1[me] / c;
            return;
        }
        float v0 = 0.0f, v1 = 0.0f;
        if (fillhole) {
            const ProjScan l = proj_walk_plain(pc, (int64_t)y * g.w, 1, x, g.w, -1), r = proj_walk_plain(pc, (int64_t)y * g.w, 1, x, g.w, +1);
            const ProjScan u = proj_walk_plain(pc, x, g.w, y, g.h, -1), d = proj_walk_plain(pc, x, g.w, y, g.h, +1);
            if (l.cnt + r.cnt + u.cnt + d.cnt > 0.0f) {
                const float lt = (l.cnt > 0.0f) ? 1.0f : 0.0f, rt = (r.cnt > 0.0f) ? 1.0f : 0.0f;
                const float ut = (u.cnt > 0.0f) ? 1.0f : 0.0f, dt = (d.cnt > 0.0f) ? 1.0f : 0.0f;
                const float den = lt + rt + ut + dt;
                const int64_t il = (int64_t)y * g.w + l.pos, ir = (int64_t)y * g.w + r.pos;
                const int64_t iu = (int64_t)u.pos * g.w + x, id = (int64_t)d.pos * g.w + x;
                // a neighbour found by a walk has count > 0; the others carry weight 0 (their cell is a hole: value 0)
                const float a0 = l.cnt > 0.0f ? p0[il] / l.cnt : 0.0f, b0 = r.cnt > 0.0f ? p0[ir] / r.cnt : 0.0f;
                const float c0 = u.cnt > 0.0f ? p0[iu] / u.cnt : 0.0f, d0 = d.cnt > 0.0f ? p0[id] / d.cnt : 0.0f;
                const float a1 = l.cnt > 0.0f ? p1[il] / l.cnt : 0.0f, b1 = r.cnt > 0.0f ? p1[ir] / r.cnt : 0.0f;
                const float c1 = u.cnt > 0.0f ? p1[iu] / u.cnt : 0.0f, d1 = d.cnt > 0.0f ? p1[id] / d.cnt : 0.0f;
                v0 = (lt * a0 + rt * b0 + ut * c0 + dt * d0) / den;
                v1 = (lt * a1 + rt * b1 + ut * c1 + dt * d1) / den;
            }
        }
        o0[row + x] = v0;
        o1[row + x] = v1;
        return;
    }
    if (!fillhole) return;
    const float* cn = count + (int64_t)b * sc.b;
    if (!(cn[(int64_t)y * sc.h + x] <= 0.0f)) return;
    // B ran its normal path and left the bitmaps
    const int* rl = bits + g.rowmap + (b * g.h + y) * g.rmw;
    const int* cl = bits + g.colmap + (b * g.w + x) * g.cmw;
    const int xl = proj_bit_walk(rl, x, g.w, -1), xr = proj_bit_walk(rl, x, g.w, +1);
    const int yu = proj_bit_walk(cl, y, g.h, -1), yd = proj_bit_walk(cl, y, g.h, +1);
    // a walk that found nothing contributes weight 0; its position only has to be valid
    ProjScan l, r, u, d;
    l.pos = xl < 0 ? x : xl; r.pos = xr < 0 ? x : xr; u.pos = yu < 0 ? y : yu; d.pos = yd < 0 ? y : yd;
    l.cnt = xl < 0 ? 0.0f : cn[(int64_t)y * sc.h + xl];
    r.cnt = xr < 0 ? 0.0f : cn[(int64_t)y * sc.h + xr];
    u.cnt = yu < 0 ? 0.0f : cn[(int64_t)yu * sc.h + x];
    d.cnt = yd < 0 ? 0.0f : cn[(int64_t)yd * sc.h + x];
    if (l.cnt + r.cnt + u.cnt + d.cnt <= 0.0f) return;
    const float lt = (l.cnt > 0.0f) ? 1.0f : 0.0f;
    const float rt = (r.cnt > 0.0f) ? 1.0f : 0.0f;
    const float ut = (u.cnt > 0.0f) ? 1.0f : 0.0f;
    const float dt = (d.cnt > 0.0f) ? 1.0f : 0.0f;
    const float den = lt + rt + ut + dt;
    o0[row + x] = (lt * o0[row + l.pos] + rt * o0[row + r.pos] + ut * o0[(int64_t)u.pos * s1.h + x] +
                   dt * o0[(int64_t)d.pos * s1.h + x]) / den;
    o1[row + x] = (lt * o1[row + l.pos] + rt * o1[row + r.pos] + ut * o1[(int64_t)u.pos * s1.h + x] +
                   dt * o1[(int64_t)d.pos * s1.h + x]) / den;
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

// Per (device, stream) workspace for the tile lists, allocated on first use and grown on demand.
// hipMalloc happens on the first call for a given stream / a larger frame only (do a warm-up call
// before capturing into a graph); calls on one stream are ordered, so one workspace per stream is
// enough and two streams never share one.
struct ProjWorkspace {
    int device; hipStream_t stream;
    int* words; size_t capacity;
    int* bits; size_t bit_capacity;
    float* planes; size_t plane_capacity;       // fallback scratch: 3 dense planes, zero between calls
    int serial;
};
static std::mutex g_ws_mutex;
static std::deque<ProjWorkspace> g_ws;            // deque: a record handed to one caller stays put when another adds one

static ProjWorkspace* proj_workspace(hipStream_t st, size_t words, size_t bit_words, size_t plane_floats) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    ProjWorkspace* w = nullptr;
    for (auto& e : g_ws)
        if (e.device == dev && e.stream == st) w = &e;
    if (!w) {
        g_ws.push_back(ProjWorkspace{dev, st, nullptr, 0, nullptr, 0, nullptr, 0, 0});
        w = &g_ws.back();
    }
    if (w->capacity < words) {
        if (w->words) (void)hipFree(w->words);              // synchronises: no kernel still uses it
        w->words = nullptr;
        w->capacity = 0;
        if (hipMalloc(&w->words, words * sizeof(int)) != hipSuccess) return nullptr;
        if (hipMemset(w->words, 0, words * sizeof(int)) != hipSuccess) return nullptr;
        w->capacity = words;
        w->serial = 0;
        // the "dirty" word went with the old header: make the scratch planes clean by hand
        if (w->planes && hipMemset(w->planes, 0, w->plane_capacity * sizeof(float)) != hipSuccess) return nullptr;
    }
    if (w->bit_capacity < bit_words) {
        if (w->bits) (void)hipFree(w->bits);
        w->bits = nullptr;
        w->bit_capacity = 0;
        if (hipMalloc(&w->bits, bit_words * sizeof(int)) != hipSuccess) return nullptr;
        w->bit_capacity = bit_words;                        // A clears what a call uses
    }
    if (w->plane_capacity < plane_floats) {
        if (w->planes) (void)hipFree(w->planes);
        w->planes = nullptr;
        w->plane_capacity = 0;
        if (hipMalloc(&w->planes, plane_floats * sizeof(float)) != hipSuccess) return nullptr;
        if (hipMemset(w->planes, 0, plane_floats * sizeof(float)) != hipSuccess) return nullptr;
        w->plane_capacity = plane_floats;
    }
    w->serial += 1;                                         // serial 0 never matches: the header starts at 0
    return w;
}

// s1 = strides of `out` (the reference binding shares them with the input flow)
template <bool DEPTH, bool UP>
static int project_forward(const ProjFlow& flow, const float* in2, float* count, float* out, int batch, int h, int w,
                           int fillhole, vfi_strides s1, vfi_strides s2, vfi_strides sc, hipStream_t st) {
    ProjGeom g;
    g.h = h; g.w = w;
    g.tiles_x = (w + PROJ_TW - 1) / PROJ_TW;
    g.tiles_y = (h + PROJ_TH - 1) / PROJ_TH;
    const int64_t nt = (int64_t)g.tiles_x * g.tiles_y * batch;
    if (nt > (1 << 24) || g.tiles_x > (1 << PROJ_SEG_SHIFT) || (int64_t)batch * h >= (1 << (31 - PROJ_SEG_SHIFT)))
        return VFI_ERR_SHAPE;
    g.ntiles = (int)nt;
    g.rmw = (w + 31) / 32;
    g.cmw = (g.tiles_y * PROJ_TH + 31) / 32;                // whole tiles: B ORs 16-bit halves
    const size_t tile_words = proj_ws_tile_words(g.ntiles);
    const size_t bit_words = (size_t)batch * ((size_t)h * g.rmw + (size_t)w * g.cmw);
    if (bit_words > (size_t)INT_MAX) return VFI_ERR_SHAPE;
    g.rowmap = 0;
    g.colmap = batch * h * g.rmw;
    ProjWorkspace* ws = proj_workspace(st, tile_words, bit_words, (size_t)3 * batch * h * w);
    if (!ws) return VFI_ERR_LAUNCH;
    hipLaunchKernelGGL((proj_bin<DEPTH, UP>), dim3(g.ntiles), dim3(PROJ_THREADS), 0, st, flow, in2, g, s2, ws->words,
                       ws->bits, ws->planes, (int64_t)ws->plane_capacity, ws->serial);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    hipLaunchKernelGGL((proj_gather<DEPTH, UP>), dim3(g.ntiles), dim3(PROJ_THREADS), 0, st, flow, in2, count, out, g, s1,
                       s2, sc, ws->words, ws->bits, ws->planes, ws->serial);
    if (launch_status() != VFI_OK) return VFI_ERR_LAUNCH;
    // (also runs with fillhole == 0: it is where the fallback path normalises)
    hipLaunchKernelGGL(proj_finish, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, st, count, out, g, s1, sc,
                       ws->words, ws->bits, ws->planes, fillhole, ws->serial);
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
