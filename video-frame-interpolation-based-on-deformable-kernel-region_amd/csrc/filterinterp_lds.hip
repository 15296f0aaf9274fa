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
// move as full 256-B rows; every thread owns FI_PX pixels of one column), finds
// the bounding box of all its taps from the flow, and per channel stages exactly
// that window of the image plane into LDS with coalesced row reads (borders
// replicated while staging, so a pixel's 4x4 window is always 4 contiguous floats
// x 4 rows in LDS).  The 16 taps then are LDS reads at immediate offsets.  Flow,
// blend weights, the 16 filter taps and the LDS window address stay in registers
// for all channels.
//
// Pipeline: the op streams every image plane once, so it lives on memory-level
// parallelism.  Windows are staged by LDS-DMA (global_load_lds_dword: global ->
// LDS, no staging registers) into a ring of R slots carved from one 64 KB LDS
// array; while channel c is computed, the windows of channels c+1 .. c+R-1 are in
// flight or landed.  Per channel: compute, issue the DMA of channel c+R-1 into
// the slot freed last iteration, counted s_waitcnt vmcnt(N) for channel c+1 only,
// one raw s_barrier.  R-1 windows in flight per workgroup, two workgroups per CU.
//
// A tile whose tap window does not fit the LDS budget (wildly divergent flow)
// gathers from global memory instead -- same results, decided per workgroup.
//
// Launch: 1-D grid of tiles in row-major order (an XCD-contiguous band mapping
// was measured slower, see the kernel).  An optional split of the channel range
// over blockIdx.y shortens the tail when the tile count does not fill the chip
// evenly.
#include "filterinterp_dev.h"

#include <limits.h>

#include <type_traits>

namespace vfi {

// compile-time loop: the body sees a constant index, so register arrays indexed by it stay in
// registers (a runtime-indexed array would be demoted to scratch)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

#define FI_TW 64
#ifndef FI_TH
#define FI_TH 16
#endif
#ifndef FI_PX
#define FI_PX 2                                     // pixels per thread (rows y, y + FI_TH/FI_PX, ...)
#endif
#define FI_THREADS (FI_TW * FI_TH / FI_PX)          // 512
#define FI_PASS_ROWS (FI_TH / FI_PX)
#define FI_KS (FI_PX / 2)                           // staged elements per thread scale with the pixels per thread
#define FI_HDR 16                                   // floats at the head of the LDS array (bounding box)
#ifndef FI_RING_FLOATS
#define FI_RING_FLOATS 20464                        // LDS ring: 20480 floats = 81,920 B with the header: two workgroups = a CU's 160 KB (round 4; 64,000 B before)
#endif
#ifndef FI_RMAX
#define FI_RMAX 5                                   // ring slots, at most (4 windows in flight)
#endif
#define FI_KTOP ((FI_RING_FLOATS / (2 * FI_THREADS)) < 15 * FI_KS ? 12 * FI_KS : 15 * FI_KS)    // staged elements per thread and channel, at most (two ring slots)
#define FI_XCDS 8
#define FI_B64_MIN_BH 34                             // bounding box from which a tile takes the aligned 8-byte tap reads
#define FI_B64_MIN_BW 92
#ifdef VFI_DEV
#define FI_ABL(flags) (((flags) >> 20) & 255)       // development: parts of the lean loop switched off or aliased (wrong results, timing only)
#else
#define FI_ABL(flags) 0
#endif

typedef __attribute__((address_space(3))) void* fi_lptr_t;

#ifdef FI_STAMPS            // development build only: where a channel step's cycles go (tools/fm_stamps.py --single)
__device__ unsigned long long g_fi_stamps[8];       // s_memtime ticks: [0] staging issue, [1] compute, [2] vmcnt wait, [3] barrier, [4] steps
#define FI_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define FI_T(v)
#endif

__device__ __forceinline__ int wave_min(int v) { return wave_min_i32(v); }
__device__ __forceinline__ int wave_max(int v) { return wave_max_i32(v); }

struct FiWindow { int bx0, by0, bw, bh, pitch, h, w, hs; };
struct FiPixel {
    bool valid, inimg;
    float alpha, beta;
    int lbase;              // LDS index of the pixel's 4x4 window origin inside a staged window (B64: rounded down to even)
    bool odd;               // B64: the origin's column inside the window is odd
    unsigned pix;           // element offset of the pixel inside an image plane
    float f[16];
};
// Optional epilogue (DAIN.FilterInterpolate, networks/DAIN.py:560-573 / DAIN_slowmotion.py:324-335): this launch is the
// second of the pair; besides its own result v it writes blend = other * w0 + v * w2 (products rounded separately, as
// torch's three elementwise ops), `other` being the first launch's output.  For frames only (FI_BLEND_MAXC channels at
// most): the partner value is loaded inside the channel loop, which would cost a deep ring its depth.
#define FI_BLEND_MAXC 4
struct FiBlend { const float* other; float* out; float w0, w2; };

// s_waitcnt vmcnt(G*K): everything but the youngest G staged windows (K DMA loads each) has landed.
// The pixel stores of the compute phases sit in the same in-order counter; not counting them
// only makes the wait stricter.
template <int K>
__device__ __forceinline__ void fi_wait_windows(int younger_groups) {
    switch (younger_groups) {
    case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K < 63 ? K : 63) : "memory"); break;
    case 2:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K < 63 ? 2 * K : 63) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * K < 63 ? 3 * K : 63) : "memory"); break;
    }
}

// Channel loop of one workgroup: K staged elements per thread and channel, ring of R slots.
template <int K, bool BLEND>
__device__ __forceinline__ void fi_run_channels(const float* __restrict__ img, float* __restrict__ out, int64_t cs,
                                                int c_begin, int c_end, int tid, const FiWindow& win,
                                                const FiPixel (&px)[FI_PX], float* __restrict__ ring, int R,
                                                int flags, const FiBlend& bl) {
    static_assert((FI_RING_FLOATS / (K * FI_THREADS) < FI_RMAX ? FI_RING_FLOATS / (K * FI_THREADS) - 2 : FI_RMAX - 2) * K <= 63, "vmcnt is a 6-bit counter");
    // Element e = tid + k*FI_THREADS of the staged window, row-major with row pitch `pitch` = bw
    // rounded up to a multiple of 32 floats: with the pitch a multiple of the 32 LDS banks a tap's
    // bank depends on its column only, so lanes of a wave whose windows sit on different rows do
    // not collide (measured with the compact pitch bw: 58 % of the LDS cycles were conflicts).
    // Pad columns and elements past the last row get an out-of-range buffer offset: the load
    // returns 0 without touching memory, so the DMA loads are unconditional straight-line code.
    // Addressing is buffer-style: a wave-uniform descriptor of the channel's plane (rebuilt per
    // channel from scalars) + one 32-bit byte offset per element that never changes -- no
    // per-channel vector address arithmetic, one VGPR per element.  One buffer_load_dword ... lds
    // writes 64 consecutive floats: LDS destination = wave-uniform base + lane*4 = this layout.
    const float inv_pitch32 = 1.0f / (float)win.pitch;
    unsigned goff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + k * FI_THREADS;
        const int r = fi_row_of(e, inv_pitch32);
        const int col = e - r * win.pitch;
        const unsigned off = 4u * (unsigned)(clampi(win.by0 + r, 0, win.h - 1) * win.hs + clampi(win.bx0 + col, 0, win.w - 1));
        goff[k] = (col < win.bw && r < win.bh) ? off : 0x80000000u;
    }
    const int plane_bytes = 4 * ((win.h - 1) * win.hs + win.w);
    constexpr int NP = K * FI_THREADS;                      // floats per ring slot
    const int D = R - 1;                                    // windows in flight
    auto issue = [&](int c, int slot) {
        const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)c * cs), 0, plane_bytes, 0x00020000);
        // (the LDS destination is M0 = the wave's first element.  Formed on the scalar unit from a provably uniform wave id it
        //  would save a vector add and a v_readfirstlane per load -- but hipcc then sees the DMA target alias the tap reads
        //  and drains vmcnt before every LDS read: 1.05 -> 1.61 ms.  From tid it does not.)
        float* l = ring + slot * NP + tid;
#pragma unroll
        for (int k = 0; k < K; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (fi_lptr_t)(l + k * FI_THREADS), 4, goff[k], 0, 0, 0);
    };
    auto finish = [&](int c, int p, const float (&v)[16], float* o) {
        const float val = fi4_pixel(v, px[p].f, px[p].alpha, px[p].beta);
        o[px[p].pix] = val;
        if constexpr (BLEND) {
            // (a load inside the counted-vmcnt loop drains the ring -- harmless here: a frame's three windows
            //  were all issued before the loop)
            const float* bi = bl.other + (int64_t)c * cs;
            const float q0 = bi[px[p].pix] * bl.w0, q2 = val * bl.w2;
            float* bo = bl.out + (int64_t)c * cs;               // (wave-uniform plane pointer + 32-bit pixel offset)
            bo[px[p].pix] = q0 + q2;
        }
    };
    auto compute = [&](int c, int slot) {
        float* o = out + (int64_t)c * cs;
        const float* base = ring + slot * NP;
        {
#pragma unroll
            for (int p = 0; p < FI_PX; ++p) {
                if (px[p].valid) {
                    const float* t = base + px[p].lbase;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[r * 4 + k] = t[r * win.pitch + k];
                    finish(c, p, v, o);
                }
            }
        }
    };

    if (c_begin >= c_end) return;
    const int last = c_end - 1;
    for (int j = 0; j < D; ++j)
        if (c_begin + j <= last) issue(c_begin + j, j);
    fi_wait_windows<K>(min(c_begin + D - 1, last) - c_begin);      // window of c_begin has landed ...
    __builtin_amdgcn_s_barrier();                                   // ... in every wave
    int slot = 0;
    for (int c = c_begin; c <= last; ++c) {
        // the slot read in the previous iteration is free: every wave passed that barrier
        // (issuing before the compute phase measured ~5 % faster than after it: flag bit 0)
        if (!(flags & 1) && c + D <= last) issue(c + D, slot == 0 ? R - 1 : slot - 1);
        compute(c, slot);
        if ((flags & 1) && c + D <= last) issue(c + D, slot == 0 ? R - 1 : slot - 1);
        if (c < last) fi_wait_windows<K>(min(c + D, last) - (c + 1));
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    // copy-through of the (rare) invalid pixels (:2814-2818), kept out of the pipelined loop so
    // that the loop's only vector-memory loads are the staged windows
#pragma unroll
    for (int p = 0; p < FI_PX; ++p)
        if (px[p].inimg && !px[p].valid)
            for (int c = c_begin; c < c_end; ++c) {
                const float val = img[(int64_t)c * cs + px[p].pix];
                out[(int64_t)c * cs + px[p].pix] = val;
                if constexpr (BLEND) {
                    const float q0 = bl.other[(int64_t)c * cs + px[p].pix] * bl.w0, q2 = val * bl.w2;
                    bl.out[(int64_t)c * cs + px[p].pix] = q0 + q2;
                }
            }
}

// The same channel loop written for the fewest instructions per pixel.  Counters and ablations say what bounds the
// plain loop is instruction issue, not bytes: a launch with neither window staging nor result stores still takes 0.88 of
// 1.05 ms, SQ_ACTIVE_INST_ANY x 4 waves fills a SIMD's cycles, 38 % fewer LDS cycles (8-byte reads) or overlapping
// the LDS reads with the arithmetic change nothing, and a wave of the plain loop issues ~175 instructions per channel for
// 2 x (8 tap reads + 21 multiply-adds + 1 store).  Here:
//  * the ring geometry is a compile-time function of K: the steady-state wait is ONE s_waitcnt with a constant
//    (the plain loop recomputes min(c + D, last) - (c + 1) and branches four ways every channel); the last D channels run
//    in a second loop that waits for everything;
//  * plane descriptors advance by one plane per channel (two scalar adds) instead of a 64-bit multiply;
//  * the DMA destination M0 comes from a wave id the compiler can see is uniform: one scalar add per load instead of a
//    vector add + v_readfirstlane + s_mov (safe only because the tap reads are asm: see issue() in the plain loop);
//  * tap reads are asm with two hand-placed lgkmcnt waits per channel instead of fourteen, and the second pixel's first
//    reads are in flight while the first pixel is multiplied;
//  * results leave through buffer stores whose offset is out of range for an invalid pixel: no exec-mask juggling.
//  * DMA16 (built in round 3, on since round 4): a window whose columns all lie inside the image, in a tensor whose rows are
//    16-byte aligned, starts on a multiple of four columns and is staged in 16-byte units (K counts units per thread): a
//    quarter of the DMA instructions.  Same window contents, same tap reads, same bits.  With the 64,000-byte ring of round 3
//    the coarser units cost two ring slots and the launch 4-8 %; with 81,920 bytes (K = 2 units: four slots) it gains
//    1.5-3 % on the smooth field and 5 % on the quarter field (tools/fi_variant_check.py --flags 0x8,0x10000008).
template <int K, bool B64, bool DMA16 = false>
__device__ __forceinline__ void fi_run_channels_lean(const float* __restrict__ img, float* __restrict__ out, int64_t cs,
                                                     int c_begin, int c_end, int tid, const FiWindow& win,
                                                     const FiPixel (&px)[FI_PX], float* __restrict__ ring, int abl) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr int EPT = DMA16 ? 4 : 1;                      // floats per staged element
    constexpr int NP = K * FI_THREADS * EPT;
    constexpr int R = (FI_RING_FLOATS / NP) < FI_RMAX ? (FI_RING_FLOATS / NP) : FI_RMAX;
    constexpr int D = R - 1;
    static_assert(D >= 1 && (D - 1) * K <= 63, "ring geometry");
    if (c_begin >= c_end) return;
    const float inv_pitch32 = 1.0f / (float)win.pitch;
    unsigned goff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = (tid + k * FI_THREADS) * EPT;
        const int r = fi_row_of(e, inv_pitch32);
        const int col = e - r * win.pitch;
        // (DMA16: bx0 is a multiple of 4 and bx0 .. bx0 + bw - 1 lie inside the image, whose width is a multiple of 4)
        const unsigned off = 4u * (unsigned)(clampi(win.by0 + r, 0, win.h - 1) * win.hs + (DMA16 ? win.bx0 + col : clampi(win.bx0 + col, 0, win.w - 1)));
        goff[k] = (col < win.bw && r < win.bh) ? off : 0x80000000u;
    }
    const int plane_bytes = 4 * ((win.h - 1) * win.hs + win.w);
    const int wave_first = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
    const unsigned ring_lds = (unsigned)(uintptr_t)(fi_lptr_t)ring;
    const unsigned pitch4 = 4u * (unsigned)win.pitch;
    unsigned lb[FI_PX], soff[FI_PX];
#pragma unroll
    for (int p = 0; p < FI_PX; ++p) {
        lb[p] = ring_lds + 4u * (unsigned)px[p].lbase;          // (an invalid pixel's reads land past the LDS: they return 0)
        soff[p] = px[p].valid ? 4u * px[p].pix : 0x80000000u;   // (and its store is dropped by the range check)
    }
    // filter taps as (left quadrant, right quadrant) pairs: rows 0-1 feed the top sums, rows 2-3 the bottom ones
    v2f F[FI_PX][8];
#pragma unroll
    for (int p = 0; p < FI_PX; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            F[p][2 * r] = v2f{px[p].f[4 * r], px[p].f[4 * r + 2]};
            F[p][2 * r + 1] = v2f{px[p].f[4 * r + 1], px[p].f[4 * r + 3]};
        }
    const int last = c_end - 1;
    const float* pdma = img + (int64_t)c_begin * cs;            // plane the next window is staged from
    float* pout = out + (int64_t)c_begin * cs;                  // plane the next results go to
    int alias_in = 0, alias_out = 0;                            // (development: abl bits 6 / 7 keep the planes read / written inside 16)
    // 16-byte-staging instances: ONE descriptor per tensor, the plane in the instruction's scalar offset -- one scalar add per
    // channel and tensor instead of a 64-bit pointer add and a descriptor rebuild (a fifth of the loop's scalar instructions;
    // 2 % of the C=196 launch).  The descriptor spans 2^31 - 1 bytes from its base; when the next plane would end beyond that
    // the base moves up to the current plane (1080p: never; 4K: every 64th channel).  The range check compares the vector
    // offset with num_records - soffset: a lane whose offset is 0x80000000 is dropped whatever the plane.  (Host: bit 29
    // promises 4 * cs + plane_bytes < 2^31.)
    constexpr bool ONE = DMA16;
    auto din = __builtin_amdgcn_make_buffer_rsrc((void*)pdma, 0, 0x7fffffff, 0x00020000);
    auto dout = __builtin_amdgcn_make_buffer_rsrc((void*)pout, 0, 0x7fffffff, 0x00020000);
    const int cs4 = (int)(cs * 4);
    const int soff_max = 0x7fffffff - plane_bytes - cs4;      // largest scalar offset whose successor plane is still in range
    int sin = 0, sout = 0;
    auto next_in = [&]() {
        if (__builtin_expect(sin > soff_max, 0)) {
            asm volatile("; the input descriptor moves up" ::: "memory");    // (keeps this a branch: as selects it costs ten scalar instructions per channel)
            pdma = (const float*)((const char*)pdma + sin);
            din = __builtin_amdgcn_make_buffer_rsrc((void*)pdma, 0, 0x7fffffff, 0x00020000);
            sin = 0;
        }
        sin += cs4;
        if ((abl & 64) && (++alias_in & 15) == 0) sin -= 16 * cs4;
    };
    auto next_out = [&]() {                                    // (the skewed loop still stores to the plane before: offset sout - cs4 >= 0)
        if (__builtin_expect(sout > soff_max, 0)) {
            asm volatile("; the output descriptor moves up" ::: "memory");
            pout = (float*)((char*)pout + sout);
            dout = __builtin_amdgcn_make_buffer_rsrc((void*)pout, 0, 0x7fffffff, 0x00020000);
            sout = 0;
        }
        sout += cs4;
        if ((abl & 128) && ++alias_out == 16) { alias_out = 1; sout -= 15 * cs4; }
    };
    constexpr unsigned SLOT = NP * 4, RING = R * SLOT;       // bytes
    auto issue = [&](unsigned slot) {                           // (slots by their byte offset in the ring: one scalar add per step instead of a multiply)
        const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)pdma, 0, plane_bytes, 0x00020000);
        float* l = ring + (slot >> 2) + wave_first * EPT;
        if (!(abl & 2)) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (ONE) __builtin_amdgcn_raw_ptr_buffer_load_lds(din, (fi_lptr_t)(l + k * FI_THREADS * EPT), 4 * EPT, goff[k], sin, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (fi_lptr_t)(l + k * FI_THREADS * EPT), 4 * EPT, goff[k], 0, 0, 0);
            }
        }
        if constexpr (ONE) { next_in(); return; }
        pdma += cs;
        if ((abl & 64) && (++alias_in & 15) == 0) pdma -= 16 * cs;
    };
#define FI_READ2(dst, addr, o0, o1) asm volatile("ds_read2_b32 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(dst) : "v"(addr))
#define FI_READ64(dst, addr, o) asm volatile("ds_read_b64 %0, %1 offset:" #o : "=v"(dst) : "v"(addr))
    auto compute = [&](unsigned so) {
        const auto oplane = __builtin_amdgcn_make_buffer_rsrc((void*)pout, 0, plane_bytes, 0x00020000);
        // Tap reads of pixel p come in two parts (4-byte reads: rows 0-1, then rows 2-3; 8-byte reads: row 0, then rows 1-3);
        // before pixel p is multiplied, all of its reads and the first part of pixel p + 1's have been issued: two pixels'
        // registers ping-pong, at most 12 (15) LDS reads are outstanding.
        constexpr int NQ = B64 ? 12 : 8;                    // register pairs per pixel
        constexpr int PART0 = B64 ? 3 : 4;                  // reads in the first part
        v2f q[2][NQ];
        if (abl & 4) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) { q[0][i] = v2f{(float)i, 1.0f}; q[1][i] = v2f{2.0f, (float)i}; }
        }
        auto reads = [&](auto P, auto H) {
            constexpr int p = decltype(P)::value, h = decltype(H)::value;
            if (abl & 4) return;
            v2f (&d)[NQ] = q[p & 1];
            if constexpr (B64) {
                if constexpr (h == 0) {
                    const unsigned a = lb[p] + so;
                    FI_READ64(d[0], a, 0); FI_READ64(d[1], a, 8); FI_READ64(d[2], a, 16);
                } else {
                    unsigned a = lb[p] + so;
#pragma unroll
                    for (int r = 1; r < 4; ++r) { a += pitch4; FI_READ64(d[3 * r], a, 0); FI_READ64(d[3 * r + 1], a, 8); FI_READ64(d[3 * r + 2], a, 16); }
                }
            } else {
                // a read fetches columns (0, 2) or (1, 3) of a tap row: the two halves of a register pair then belong to the
                // left and the right quadrant, and one packed multiply-add advances both quadrant sums
                unsigned a = lb[p] + so + (h ? 2u * pitch4 : 0u);
                FI_READ2(d[4 * h], a, 0, 2); FI_READ2(d[4 * h + 1], a, 1, 3);
                a += pitch4;
                FI_READ2(d[4 * h + 2], a, 0, 2); FI_READ2(d[4 * h + 3], a, 1, 3);
            }
        };
        auto arrived = [&](auto P) {        // waits until pixel p's reads are back: what may stay outstanding is the part issued after them
            constexpr int p = decltype(P)::value;
            constexpr int later = (p + 1 < FI_PX) ? PART0 : 0;
            v2f (&d)[NQ] = q[p & 1];
            if constexpr (B64)
                asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                                                         "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]) : "n"(later));
            else
                asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                                                        "+v"(d[6]), "+v"(d[7]) : "n"(later));
        };
        auto pixel = [&](auto P) {
            constexpr int p = decltype(P)::value;
            const v2f (&d)[NQ] = q[p & 1];
            float val;
            if constexpr (B64) {
                const bool odd = px[p].odd;
                float v[16];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r * 4 + 0] = odd ? d[3 * r].y : d[3 * r].x;
                    v[r * 4 + 1] = odd ? d[3 * r + 1].x : d[3 * r].y;
                    v[r * 4 + 2] = odd ? d[3 * r + 1].y : d[3 * r + 1].x;
                    v[r * 4 + 3] = odd ? d[3 * r + 2].x : d[3 * r + 1].y;
                }
                val = fi4_pixel(v, px[p].f, px[p].alpha, px[p].beta);
            } else {
                v2f top = d[0] * F[p][0];                   // (same order per quadrant sum as fi4_pixel)
                top = __builtin_elementwise_fma(d[1], F[p][1], top);
                top = __builtin_elementwise_fma(d[2], F[p][2], top);
                top = __builtin_elementwise_fma(d[3], F[p][3], top);
                v2f bot = d[4] * F[p][4];
                bot = __builtin_elementwise_fma(d[5], F[p][5], bot);
                bot = __builtin_elementwise_fma(d[6], F[p][6], bot);
                bot = __builtin_elementwise_fma(d[7], F[p][7], bot);
                val = blend4(px[p].alpha, px[p].beta, top.x, top.y, bot.x, bot.y);
            }
            if (!(abl & 1) || val == 123456.789f) {
                if constexpr (ONE) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), dout, soff[p], sout, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), oplane, soff[p], 0, 0);
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        reads(I0{}, I0{}); reads(I0{}, I1{});
        if constexpr (FI_PX > 1) reads(I1{}, I0{});
        static_for<0, FI_PX>([&](auto P) {
            constexpr int p = decltype(P)::value;
            arrived(P);
            pixel(P);
            if constexpr (p + 1 < FI_PX) reads(std::integral_constant<int, p + 1>{}, I1{});
            if constexpr (p + 2 < FI_PX) reads(std::integral_constant<int, p + 2>{}, I0{});
        });
        if constexpr (ONE) { next_out(); return; }
        pout += cs;
        if ((abl & 128) && ++alias_out == 16) { alias_out = 1; pout -= 15 * cs; }    // (planes 1 .. 15 after the first lap: the skewed store reaches one plane back)
    };
    // Two pixels, 4-byte reads: the pipeline is skewed by one pixel across the barrier.  In channel c a wave issues pixel 0's
    // reads, multiplies pixel 1 of channel c - 1 (its taps were read before the barrier and wait in registers), issues pixel
    // 1's reads, multiplies pixel 0, and waits for pixel 1's taps: every LDS read is in flight under arithmetic of the same
    // wave, none is waited for with nothing to do (the plain order exposes the first reads after each barrier).
    constexpr bool SKEW = !B64 && FI_PX == 2 && K <= 7 * FI_KS;      // (4-byte reads; with 8-byte reads -- 24 more registers in flight -- it measured 30 % slower; the four largest ring geometries have no registers to spare: beside the 16-byte-staging instances the K = 10 loop, then the K = 8 loop reloaded registers inside their counted pipelines)
    constexpr int NQS = B64 ? 12 : 8;                        // register pairs per pixel
    v2f qa[NQS], qb[NQS];
    auto rd_all = [&](v2f (&d)[NQS], int p, unsigned so) {
        unsigned a = lb[p] + so;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (B64) { FI_READ64(d[3 * r], a, 0); FI_READ64(d[3 * r + 1], a, 8); FI_READ64(d[3 * r + 2], a, 16); }
            else { FI_READ2(d[2 * r], a, 0, 2); FI_READ2(d[2 * r + 1], a, 1, 3); }
            a += pitch4;
        }
    };
    auto arrived_s = [&](v2f (&d)[NQS], auto LATER) {       // waits until all but the `later` youngest LDS reads are back
        constexpr int later = decltype(LATER)::value;
        if constexpr (B64)
            asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                                                     "+v"(d[6]), "+v"(d[7]), "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]) : "n"(later));
        else
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                                                    "+v"(d[6]), "+v"(d[7]) : "n"(later));
    };
    auto fma_store = [&](const v2f (&d)[NQS], int p, const float* plane_ptr) {
        float val;
        if constexpr (B64) {
            const bool odd = px[p].odd;
            float v[16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r * 4 + 0] = odd ? d[3 * r].y : d[3 * r].x;
                v[r * 4 + 1] = odd ? d[3 * r + 1].x : d[3 * r].y;
                v[r * 4 + 2] = odd ? d[3 * r + 1].y : d[3 * r + 1].x;
                v[r * 4 + 3] = odd ? d[3 * r + 2].x : d[3 * r + 1].y;
            }
            val = fi4_pixel(v, px[p].f, px[p].alpha, px[p].beta);
        } else {
            v2f top = d[0] * F[p][0];                       // (same order per quadrant sum as fi4_pixel)
            top = __builtin_elementwise_fma(d[1], F[p][1], top);
            top = __builtin_elementwise_fma(d[2], F[p][2], top);
            top = __builtin_elementwise_fma(d[3], F[p][3], top);
            v2f bot = d[4] * F[p][4];
            bot = __builtin_elementwise_fma(d[5], F[p][5], bot);
            bot = __builtin_elementwise_fma(d[6], F[p][6], bot);
            bot = __builtin_elementwise_fma(d[7], F[p][7], bot);
            val = blend4(px[p].alpha, px[p].beta, top.x, top.y, bot.x, bot.y);
        }
        if constexpr (ONE) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), dout, soff[p], sout - (p ? cs4 : 0), 0);
        } else {
            const auto oplane = __builtin_amdgcn_make_buffer_rsrc((void*)plane_ptr, 0, plane_bytes, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), oplane, soff[p], 0, 0);
        }
    };
    auto compute_skewed = [&](unsigned so, bool first) {
        rd_all(qa, 0, so);
        if (!first) fma_store(qb, 1, pout - cs);            // pixel 1 of the previous channel
        // (at most 15 LDS reads outstanding)
        if constexpr (B64) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        rd_all(qb, 1, so);
        arrived_s(qa, std::integral_constant<int, NQS>{});
        fma_store(qa, 0, pout);
        // pixel 1's taps are in registers before the barrier: the slot may be overwritten after it
        arrived_s(qb, std::integral_constant<int, 0>{});
        if constexpr (ONE) { next_out(); return; }
        pout += cs;
        if ((abl & 128) && ++alias_out == 16) { alias_out = 1; pout -= 15 * cs; }    // (planes 1 .. 15 after the first lap: the skewed store reaches one plane back)
    };
    const bool skew = SKEW && !(abl & 32);
    // prologue: the first D windows
    const int n0 = min(D, c_end - c_begin);
    // (The waits count staging loads only.  vmcnt also holds the FI_PX result stores of every step, in issue order, so "all but
    //  the (D - 1) K youngest" is stricter than "window c + 1 has landed".  Round 4 counted the stores in -- (D - 1)(K + FI_PX) in the
    //  steady state, out-of-range stores behind the prologue's windows so that the first steps see the same stream, m K + (D - 1)
    //  FI_PX in the last steps -- bit-identical, and measured nothing: 1.056-1.066 against 1.063-1.080 ms.  The simple rule stays.)
    for (int j = 0; j < n0; ++j) issue((unsigned)j * SLOT);
    fi_wait_windows<K>(n0 - 1);                                 // the first window has landed ...
    __builtin_amdgcn_s_barrier();                               // ... in every wave
    int c = c_begin;
    unsigned slot = 0, freed = RING - SLOT;                      // the window being read; the slot every wave finished reading before the last barrier
    if constexpr (SKEW) if (skew) {
#ifdef FI_STAMPS
        unsigned long long acc_i = 0, acc_c = 0, acc_w = 0, acc_b = 0, acc_n = 0;
#endif
        for (; c + D <= last; ++c) {
            FI_T(t0);
            issue(freed);
            FI_T(t1);
            compute_skewed(slot, c == c_begin);
            FI_T(t2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * K) : "memory");
            FI_T(t3);
            __builtin_amdgcn_s_barrier();
#ifdef FI_STAMPS
            const unsigned long long t4 = __builtin_amdgcn_s_memtime();
            acc_i += t1 - t0; acc_c += t2 - t1; acc_w += t3 - t2; acc_b += t4 - t3; acc_n += 1;
#endif
            freed = slot; slot = (slot + SLOT == RING) ? 0u : slot + SLOT;
        }
#ifdef FI_STAMPS
        if ((tid & 63) == 0) {
            atomicAdd(&g_fi_stamps[0], acc_i); atomicAdd(&g_fi_stamps[1], acc_c); atomicAdd(&g_fi_stamps[2], acc_w);
            atomicAdd(&g_fi_stamps[3], acc_b); atomicAdd(&g_fi_stamps[4], acc_n);
        }
#endif
        for (; c <= last; ++c) {
            compute_skewed(slot, c == c_begin);
            if (c < last) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            freed = slot; slot = (slot + SLOT == RING) ? 0u : slot + SLOT;
        }
        fma_store(qb, 1, pout - cs);                            // pixel 1 of the last channel
    }
    for (; c + D <= last; ++c) {                                // steady state: window c + D exists
        issue(freed);
        compute(slot);
        if (!(abl & 16)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * K) : "memory");      // all but the D - 1 youngest windows: c + 1 has landed
        if (!(abl & 8)) __builtin_amdgcn_s_barrier();
        freed = slot; slot = (slot + SLOT == RING) ? 0u : slot + SLOT;
    }
    for (; c <= last; ++c) {                                    // the last D channels: nothing left to stage
        compute(slot);
        if (c < last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        freed = slot; slot = (slot + SLOT == RING) ? 0u : slot + SLOT;
    }
#undef FI_READ2
#undef FI_READ64
#pragma unroll
    for (int p = 0; p < FI_PX; ++p)                             // copy-through of the (rare) invalid pixels (:2814-2818)
        if (px[p].inimg && !px[p].valid)
            for (int cc = c_begin; cc < c_end; ++cc) out[(int64_t)cc * cs + px[p].pix] = img[(int64_t)cc * cs + px[p].pix];
}

// two 512-thread workgroups per CU (4 waves per SIMD): at most 128 VGPRs
template <bool BLEND, int MODE>
#ifndef FI_WAVES
#define FI_WAVES (8 / FI_PX)                        // waves per SIMD the kernel must fit (4: two 512-thread workgroups per CU)
#endif
__global__ __launch_bounds__(FI_THREADS, FI_WAVES) void fi_forward_ori_lds(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    float* __restrict__ out, int channel, int h, int w,
    vfi_strides s1, vfi_strides s2, vfi_strides s3,
    int tiles_x, int tiles_y, int ntiles, int per_xcd, int ch_per_group, int flags, FiBlend blend) {
    // ONE LDS array (a second __shared__ object beside an LDS-DMA target makes hipcc drain vmcnt
    // before LDS reads): 16-float header holding the bounding box, then the window ring
    __shared__ float lds[FI_HDR + FI_RING_FLOATS];
    int* box = reinterpret_cast<int*>(lds);

    // ---- block -> tile (XCD-contiguous bands)
    const int bid = blockIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2, and a tile's window rows share
    // their first and last 128-byte line with the horizontal neighbours' windows.  Default: G = 4 horizontally
    // consecutive tiles go to ONE XCD (workgroups b, b + 8, b + 16, b + 24 share an XCD and start together), so
    // three of four shared lines are L2 hits: EA read requests per C=196 launch 39.1 M -> 31.2 M (5.0 -> 4.0 GB),
    // 1-9 % less time depending on the box.  Larger departures from raster order lose more than they save:
    // XCD-contiguous bands +12 % time, 4x2 / 2x2 tile blocks per XCD -25 % reads but +15-30 % time
    // (tools/fi_xcd_exp.sh).  The other mappings remain selectable in a development build.
    int tile = (flags & 2) ? (bid % FI_XCDS) * per_xcd + bid / FI_XCDS : bid;
    if ((flags >> 2) & 3) {
        const int G = 1 << ((flags >> 2) & 3);
        const int xs = bid % FI_XCDS, k = bid / FI_XCDS;
        tile = ((k / G) * FI_XCDS + xs) * G + (k % G);
    }
    int b, tyi, txi;
    if (flags & 48) {
        // experiment: GW x GH tiles per XCD (flags bit 4: 4 x 2, bit 5: 2 x 2)
        const int GW = (flags & 16) ? 4 : 2, GH = 2, GN = GW * GH;
        const int xs = bid % FI_XCDS, k = bid / FI_XCDS;
        const int j = (k / GN) * FI_XCDS + xs, m = k % GN;
        const int gpr = (tiles_x + GW - 1) / GW, gpc = (tiles_y + GH - 1) / GH;
        b = j / (gpr * gpc);
        const int jr = j - b * (gpr * gpc);
        const int gy = jr / gpr, gx = jr - gy * gpr;
        txi = gx * GW + m % GW;
        tyi = gy * GH + m / GW;
        if (txi >= tiles_x || tyi >= tiles_y || b * tiles_x * tiles_y >= ntiles) return;
    } else {
        if (tile >= ntiles) return;                         // whole workgroup leaves together
        b = tile / (tiles_x * tiles_y);
        const int trem = tile - b * (tiles_x * tiles_y);
        tyi = trem / tiles_x; txi = trem - tyi * tiles_x;
    }
    const int c_begin = blockIdx.y * ch_per_group;
    const int c_end = min(channel, c_begin + ch_per_group);

    const int tid = threadIdx.x;
    const int x = txi * FI_TW + (tid & (FI_TW - 1));
    const int y0 = tyi * FI_TH + (tid >> 6);

    // ---- this thread's pixels: flow, validity, window origin, blend weights.  The 16 filter taps of each pixel are
    // fetched in the same round trip as its flow (they do not depend on it: a pixel the flow then declares invalid
    // has loaded them for nothing) -- one dependent HBM round trip less before the first window can be staged,
    // which is what a 3-channel launch consists of.
    FiPixel px[FI_PX];
    int L[FI_PX], T[FI_PX];
    int bx_lo = INT_MAX, by_lo = INT_MAX, bx_hi = INT_MIN, by_hi = INT_MIN;
    float fxv[FI_PX], fyv[FI_PX];
#pragma unroll
    for (int p = 0; p < FI_PX; ++p) {
        const int y = y0 + p * FI_PASS_ROWS;
        px[p].inimg = x < w && y < h;
        px[p].pix = (unsigned)(y * (int)s1.h + x);
        fxv[p] = fyv[p] = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) px[p].f[k] = 0.0f;
        if (px[p].inimg) {
            const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
            fxv[p] = flow[0];
            fyv[p] = flow[s2.c];
            const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
#pragma unroll
            for (int k = 0; k < 16; ++k) px[p].f[k] = fpx[(int64_t)k * s3.c];
        }
    }
#pragma unroll
    for (int p = 0; p < FI_PX; ++p) {
        const int y = y0 + p * FI_PASS_ROWS;
        const float fx = fxv[p], fy = fyv[p];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        px[p].valid = px[p].inimg && fi_valid(fx, fy, x2, y2, w, h);
        const int ix = px[p].valid ? (int)x2 : 0, iy = px[p].valid ? (int)y2 : 0;
        L[p] = ix - 1;                                      // ix + 1 - fs/2, fs == 4
        T[p] = iy - 1;
        px[p].alpha = x2 - (float)ix;
        px[p].beta = y2 - (float)iy;
        if (px[p].valid) {
            bx_lo = min(bx_lo, L[p]); by_lo = min(by_lo, T[p]);
            bx_hi = max(bx_hi, L[p] + 3); by_hi = max(by_hi, T[p] + 3);
        }
    }

    // ---- bounding box of every tap of the tile (unclamped window coordinates)
    if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
    __syncthreads();
    {
        const int x0 = wave_min(bx_lo), y0w = wave_min(by_lo);
        const int x1 = wave_max(bx_hi), y1 = wave_max(by_hi);
        if ((tid & 63) == 0 && x0 != INT_MAX) {
            atomicMin(&box[0], x0); atomicMin(&box[1], y0w);
            atomicMax(&box[2], x1); atomicMax(&box[3], y1);
        }
    }
    __syncthreads();
    const bool any_valid = box[0] != INT_MAX;
    // MODE 0 (the product): the lean channel loop (the plain one under a blend epilogue), tap reads chosen per tile: a tall or
    // wide bounding box marks a rough flow field, where the lanes of a wave sit on many window rows and columns and LDS bank
    // conflicts dominate -- there the aligned 8-byte reads win (C=196 on the "quarter" field: 1.77 -> 1.61 ms; all tiles on
    // them: 1.58); on a smooth field they lose 3-10 % to their 16 selects per pixel and channel, and its tiles keep the
    // 4-byte reads.  Development builds: 1 = the plain loop, 2 = 8-byte reads everywhere, 3 = 4-byte reads everywhere.
    const int raw_bh = any_valid ? box[3] - box[1] + 1 : 0;
    const int raw_bw = any_valid ? box[2] - box[0] + 1 : 0;
    constexpr bool lean = !BLEND && MODE != 1;
    // (the 8-byte layout's pitch is = 32 mod 64 floats: a window that needs more than 10 x 512 elements with it keeps the
    //  4-byte reads and their tighter pitch -- the two largest ring geometries are compiled for those only)
    // 16-byte staging (flags bit 28; bit 29, set by the host: width, strides and base of input1 are multiples of 16 bytes):
    // every column of the window inside the image, rows of the tensor 16-byte aligned; the
    // window then starts on a multiple of four columns (which is even: the 8-byte reads' parity rule holds too)
    const bool can16 = lean && MODE != 1 && (flags & (3 << 28)) == (3 << 28) && any_valid && box[0] >= 0 && box[2] < w;
    const int lo = can16 ? (box[0] & ~3) : box[0];
    const int bw64 = any_valid ? box[2] - (lo & ~1) + 1 : 0;
    const bool fits64 = ((((bw64 + 31) >> 6) << 6) + 32) * raw_bh <= 10 * FI_KS * FI_THREADS;
    const bool use64 = lean && (MODE == 2 || (MODE == 0 && fits64 && (raw_bh >= FI_B64_MIN_BH || raw_bw >= FI_B64_MIN_BW)));
    const int bx0 = (use64 && any_valid) ? (lo & ~1) : lo, by0 = box[1];       // 8-byte reads: window columns keep the image's parity
    const int bw = any_valid ? box[2] - bx0 + 1 : 0;
    const int bh = raw_bh;
    // LDS row pitch: a multiple of the 32 banks; 8-byte reads see 64 banks: = 32 mod 64
    const int pitch = use64 ? (((bw + 31) >> 6) << 6) + 32 : fi_pitch_for(bw);
    const int n = pitch * bh;                               // <= (w+33)*(h+2): fits int for any real frame

#pragma unroll
    for (int p = 0; p < FI_PX; ++p) {
        const int lc = L[p] - bx0;
        px[p].odd = use64 && (lc & 1);
        px[p].lbase = (T[p] - by0) * pitch + (use64 ? (lc & ~1) : lc);
    }

    const float* img = in1 + (int64_t)b * s1.b;
    float* dst = out + (int64_t)b * s1.b;
    const FiBlend bl{blend.other ? blend.other + (int64_t)b * s1.b : nullptr, blend.out ? blend.out + (int64_t)b * s1.b : nullptr,
                     blend.w0, blend.w2};

    const int kmax = (n + FI_THREADS - 1) / FI_THREADS;     // workgroup-uniform
    if (kmax > FI_KTOP) {
        // window too large for LDS: gather from global memory (workgroup-uniform branch)
#pragma unroll
        for (int p = 0; p < FI_PX; ++p) {
            if (px[p].valid) {
                fi4_channels_direct(img, dst + px[p].pix, c_begin, c_end, s1.c, (int)s1.h, h, w, L[p], T[p], px[p].f,
                                    px[p].alpha, px[p].beta);
            } else if (px[p].inimg) {
                for (int c = c_begin; c < c_end; ++c) dst[(int64_t)c * s1.c + px[p].pix] = img[(int64_t)c * s1.c + px[p].pix];
            }
            if (BLEND && px[p].inimg)                           // (this thread wrote dst[...] itself: it reads its own stores)
                for (int c = c_begin; c < c_end; ++c) {
                    const float q0 = bl.other[(int64_t)c * s1.c + px[p].pix] * bl.w0, q2 = dst[(int64_t)c * s1.c + px[p].pix] * bl.w2;
                    bl.out[(int64_t)c * s1.c + px[p].pix] = q0 + q2;
                }
        }
        return;
    }

    const FiWindow win{bx0, by0, bw, bh, pitch, h, w, (int)s1.h};
    float* ring = lds + FI_HDR;
#define FI_RUN(K) if constexpr (lean && MODE == 0) { \
        if constexpr ((K) <= 10 * FI_KS) { if (use64) fi_run_channels_lean<K, true>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring, FI_ABL(flags)); } \
        if (!use64) fi_run_channels_lean<K, false>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring, FI_ABL(flags)); \
    } else if constexpr (lean) fi_run_channels_lean<K, MODE == 2>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring, FI_ABL(flags)); else \
                  fi_run_channels<K, BLEND>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring, \
                                     min(((flags >> 8) & 255) ? ((flags >> 8) & 255) : FI_RMAX, FI_RING_FLOATS / ((K) * FI_THREADS)), flags, bl)
    if constexpr (lean && MODE != 1) {
        // (units of four floats: two or three per thread cover the windows of smooth fields; larger ones stay on 4-byte staging)
        const int k16 = (n + 4 * FI_THREADS - 1) / (4 * FI_THREADS);
        if (can16 && k16 <= 3) {
#define FI_RUN16(K) { if (use64) fi_run_channels_lean<K, true, true>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring, FI_ABL(flags)); \
                      else fi_run_channels_lean<K, false, true>(img, dst, s1.c, c_begin, c_end, tid, win, px, ring, FI_ABL(flags)); }
            if (k16 <= 1) FI_RUN16(1) else if (k16 == 2) FI_RUN16(2) else FI_RUN16(3)
#undef FI_RUN16
            return;
        }
    }
    if (kmax <= 2 * FI_KS) FI_RUN(2 * FI_KS);
    else if (kmax <= 3 * FI_KS) FI_RUN(3 * FI_KS);
    else if (kmax <= 4 * FI_KS) FI_RUN(4 * FI_KS);
    else if (kmax <= 5 * FI_KS) FI_RUN(5 * FI_KS);
    else if (kmax <= 6 * FI_KS) FI_RUN(6 * FI_KS);
    else if (kmax <= 7 * FI_KS) FI_RUN(7 * FI_KS);
    else if (kmax <= 8 * FI_KS) FI_RUN(8 * FI_KS);
    else if (kmax <= 10 * FI_KS) FI_RUN(10 * FI_KS);
    else if (kmax <= 12 * FI_KS) FI_RUN(12 * FI_KS);
#if FI_RING_FLOATS >= 2 * 15 * FI_KS * FI_THREADS
    else FI_RUN(15 * FI_KS);
#endif
#undef FI_RUN
}

// (explicit: with the 16-byte flavours added, hipcc 7.2 silently left the host stubs of the implicitly instantiated modes 0
//  and 3 of a development build undefined)
#define FI_INSTANCE(B, M) template __global__ void fi_forward_ori_lds<B, M>(const float* __restrict__, const float* __restrict__, \
    const float* __restrict__, float* __restrict__, int, int, int, vfi_strides, vfi_strides, vfi_strides, int, int, int, int, int, int, FiBlend)
FI_INSTANCE(true, 0);
FI_INSTANCE(false, 0);
#ifdef VFI_DEV
FI_INSTANCE(false, 1);
FI_INSTANCE(false, 2);
FI_INSTANCE(false, 3);
#endif
#undef FI_INSTANCE

}  // namespace vfi

using namespace vfi;

// Kernel flags: bit 0 issue the next DMA before / after the compute phase (plain loop); bit 1 XCD-contiguous bands of tiles;
// bits 16-17 channel loop (development builds: 1 plain, 2 / 3 lean with 8- / 4-byte tap reads everywhere); bits 20-25 parts of the lean loop
// switched off (development builds, timing only);
// bit 28 16-byte staging where a window allows it (bit 29 is set by the host: input1's rows and planes 16-byte aligned);
// bits 2-3 log2 of the tiles per XCD group (default 2: four horizontally consecutive tiles on one XCD, see the
// kernel); bits 4-5 two-dimensional groups; bits 8.. ring depth.  g_fi_groups: channel groups, 0 = chosen below.
#define FI_DEFAULT_FLAGS (8 | (1 << 28))            // four tiles per XCD group; 16-byte staging where a window allows it
VFI_KNOB(int, g_fi_flags, FI_DEFAULT_FLAGS);
VFI_KNOB(int, g_fi_groups, 0);
#ifdef VFI_DEV
extern "C" void vfi_dev_filterinterp(int flags, int groups) { g_fi_flags = flags; g_fi_groups = groups; }
#endif

// returns -1 when this path does not apply (caller uses the direct kernel)
static int forward_ori_lds(const float* input1, const float* input2, const float* input3, float* output, int batch, int channel,
                           int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_stream_t stream, FiBlend blend) {
    // byte offsets inside a plane are 32-bit in the kernel
    if ((int64_t)h * s1.h * 4 > INT_MAX) return -1;
    const int tiles_x = (w + FI_TW - 1) / FI_TW, tiles_y = (h + FI_TH - 1) / FI_TH;
    const int64_t nt = (int64_t)tiles_x * tiles_y * batch;
    if (nt > (1 << 28)) return -1;
    const int ntiles = (int)nt;
    const int per_xcd = (((ntiles + FI_XCDS - 1) / FI_XCDS) + 7) & ~7;         // (a multiple of 8: the tile-group experiments)

    // split the channel range over blockIdx.y when that shortens the tail: two workgroups per
    // CU run at a time; every extra group re-reads the flow + 16 filter planes (72 B/pixel)
    // next to 8 B/pixel/channel of image traffic
    int best_groups = fi_channel_groups(ntiles, channel, 4.3);
    if (g_fi_groups > 0) best_groups = g_fi_groups < channel ? g_fi_groups : channel;
    if (blend.out) best_groups = 1;                         // (the blend epilogue keeps a pixel's channels in one workgroup)
    const int ch_per_group = (channel + best_groups - 1) / best_groups;
    const int groups = (channel + ch_per_group - 1) / ch_per_group;

    int grid_x = per_xcd * FI_XCDS;
    if (g_fi_flags & 48) {
        const int GW = (g_fi_flags & 16) ? 4 : 2, GH = 2;
        const int ngroups = ((tiles_x + GW - 1) / GW) * ((tiles_y + GH - 1) / GH) * batch;
        grid_x = ((ngroups + FI_XCDS - 1) / FI_XCDS) * FI_XCDS * GW * GH;
    }
    const dim3 grid((unsigned)grid_x, (unsigned)groups, 1);
    // (16-byte staging: rows of input1 on 16-byte boundaries; its single-descriptor addressing: two planes within 2^31 bytes)
    const int aligned16 = !(w & 3) && !(s1.h & 3) && !(s1.c & 3) && !(s1.b & 3) && !((uintptr_t)input1 & 15) &&
                          s1.c > 0 && 4 * s1.c + 4 * ((int64_t)(h - 1) * s1.h + w) < 0x7fffffffLL;
    const int kflags = g_fi_flags | (aligned16 << 29);
    if (blend.out)
        hipLaunchKernelGGL((fi_forward_ori_lds<true, 0>), grid, dim3(FI_THREADS, 1, 1), 0, (hipStream_t)stream, input1, input2,
                           input3, output, channel, h, w, s1, s2, s3, tiles_x, tiles_y, ntiles, per_xcd, ch_per_group, kflags, blend);
#ifdef VFI_DEV
#define FI_DEV_MODE(M) hipLaunchKernelGGL((fi_forward_ori_lds<false, M>), grid, dim3(FI_THREADS, 1, 1), 0, (hipStream_t)stream, input1, input2, \
                           input3, output, channel, h, w, s1, s2, s3, tiles_x, tiles_y, ntiles, per_xcd, ch_per_group, kflags, blend)
    else if (((g_fi_flags >> 16) & 3) == 1) FI_DEV_MODE(1);   // the plain channel loop
    else if (((g_fi_flags >> 16) & 3) == 2) FI_DEV_MODE(2);   // the lean loop with 8-byte tap reads
    else if (((g_fi_flags >> 16) & 3) == 3) FI_DEV_MODE(3);   // the lean loop with 4-byte tap reads only
#undef FI_DEV_MODE
#endif
    else
        hipLaunchKernelGGL((fi_forward_ori_lds<false, 0>), grid, dim3(FI_THREADS, 1, 1), 0, (hipStream_t)stream, input1, input2,
                           input3, output, channel, h, w, s1, s2, s3, tiles_x, tiles_y, ntiles, per_xcd, ch_per_group, kflags, blend);
    return launch_status();
}

#ifdef FI_STAMPS
// reads the accumulators and clears them (synchronises)
extern "C" int vfi_dev_fi_stamps(unsigned long long* host8) {
    if (hipDeviceSynchronize() != hipSuccess) return VFI_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_fi_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return VFI_ERR_LAUNCH;
    const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fi_stamps), zero, sizeof(zero)) == hipSuccess ? VFI_OK : VFI_ERR_LAUNCH;
}
#endif

extern "C" int vfi_filterinterp_forward_ori_lds(const float* input1, const float* input2, const float* input3,
                                                 float* output, int batch, int channel, int h, int w,
                                                 vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                                 vfi_stream_t stream) {
    return forward_ori_lds(input1, input2, input3, output, batch, channel, h, w, s1, s2, s3, stream, FiBlend{nullptr, nullptr, 0.0f, 0.0f});
}

// internal: the second launch of DAIN.FilterInterpolate with the blend as its epilogue -- besides output it writes
// blend = other * w0 + output * w2; other / blend have input1's strides; channel <= FI_BLEND_MAXC.  -1: not applicable.
extern "C" int vfi_filterinterp_forward_ori_lds_blend(const float* input1, const float* input2, const float* input3,
                                                       float* output, const float* other, float* blend, float w0, float w2,
                                                       int batch, int channel, int h, int w,
                                                       vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_stream_t stream) {
    if (channel > FI_BLEND_MAXC || !other || !blend) return -1;
    return forward_ori_lds(input1, input2, input3, output, batch, channel, h, w, s1, s2, s3, stream, FiBlend{other, blend, w0, w2});
}
