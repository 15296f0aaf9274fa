// filterinterp_multi.hip -- FilterInterpolation (_ori, fs == 4) forward of ONE image with SEVERAL flows at once.
//
// DAIN_slowmotion warps the same 196-channel context tensor with the same filter once per time offset
// (networks/DAIN_slowmotion.py:167-183 calls FilterInterpolate_ctx, :311-317, for t = 0.25, 0.5, 0.75: three
// FilterInterpolationModule calls per direction that differ in the flow only).  Each of those calls streams the
// whole tensor: the single-flow kernel (filterinterp_lds.hip) sits on the HBM ceiling, because a tile's window
// rows drag in the 128-byte lines they share with the neighbouring tiles (profiles/README.md: 1.8 x the
// algorithmic bytes leave the memory side).  Here a tile stages ONE window per channel -- the bounding box of the
// taps of all NT flows; the flows of a slow-motion step are scaled copies of each other, so it is only a few
// pixels larger than each flow's own window -- and produces NT outputs from it.  Per output: a third of the window
// traffic, the same LDS reads and arithmetic.  Semantics per output: filterinterpolation_cuda_kernel.cu:2692-2823,
// bit for bit the single-flow kernel's (same taps, same order).
//
// Tiling, LDS-DMA ring, counted vmcnt and tile -> XCD grouping: exactly filterinterp_lds.hip (see there); a thread
// owns two pixels, each with NT (validity, window address, blend weights) and one set of 16 filter taps.
//
// Round 4: the window is staged as PAIRS.  With NT outputs per staged window the launch is bound by its consumer side --
// LDS tap reads and vector issue (profiles/: three times the single-flow kernel's LDS cycles for one window's staging) --
// not by memory.  A tap row of the single-flow kernel is two ds_read2_b32 (columns (0, 2) and (1, 3), so that one packed
// multiply-add advances the left and the right quadrant sum): 128 B/clk.  Here the slot holds P[r][c] = (A[r][c], A[r][c + 2])
// as one 8-byte element, so the same two register pairs are two ds_read_b64 at P[c], P[c + 1] -- 8-byte aligned for every
// c, 256 B/clk, 64 banks (a stretching flow has slack).  The LDS-DMA writes consecutive dwords, so lane 2i of a staging
// instruction fetches column i and lane 2i + 1 column i + 2: every window element is staged twice (from L2 the second
// time), which the NT outputs per window pay for.  The four blend weights of an evaluation are formed once per tile
// (blend4's own products: same bits) instead of once per channel.
#include "filterinterp_dev.h"

#include <limits.h>

#include <type_traits>

namespace vfi {

#define FM_TW 64
#define FM_TH 16
#define FM_PX 2
#define FM_THREADS (FM_TW * FM_TH / FM_PX)          // 512
#define FM_PASS_ROWS (FM_TH / FM_PX)
#define FM_HDR 16
#define FM_RING_FLOATS 20464                        // with the header: 81,920 B = half of a CU's 160 KB (two workgroups per CU)
#define FM_RMAX 5
#define FM_KTOP 15                                  // staged dwords per thread and channel, at most
#define FM_KPAIR 18                                 // ... of a window staged as pairs: classes 3 x 3 ... 3 x 6, 4 x 3, 4 x 4
#define FM_XCDS 8
#define FM_MAXT 3                                   // flows per launch (4 spills inside the channel loop at 128 registers)
#ifndef FM_GROUP
#define FM_GROUP 2                                  // flows per launch the host forms groups of (round 4: three flows go as 2 + 1 --
#endif                                              // 2.78 against 3.01 ms at 1080p, C = 196: the union window of three time offsets
                                                    // costs the three-flow launch its ring depth; that launch stays in development builds)

typedef __attribute__((address_space(3))) void* fm_lptr_t;

// compile-time loop: the body sees a constant index (register arrays indexed by it stay in registers)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct FmPtrs { const float* flow[FM_MAXT]; float* out[FM_MAXT]; };

#ifndef FM_DEEP
#define FM_DEEP 0           // 1: (two flows) a whole evaluation's tap reads per register set, 16 LDS reads in flight -- measured 3.5 % SLOWER
#endif                      //    than half evaluations (1.66 against 1.60 ms per two-flow launch at 1080p, C = 196): kept as a build switch
#ifndef FM_ABL
#define FM_ABL 0            // development: parts of the paired loop switched off (wrong results, timing only): 1 stores, 2 staging, 4 tap reads
#endif
#ifdef FM_STAMPS            // development build only: where a channel step's cycles go (tools/fm_stamps.py)
__device__ unsigned long long g_fm_stamps[8];       // s_memtime ticks: [0] staging issue, [1] compute, [2] vmcnt wait, [3] barrier, [4] steps
#define FM_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define FM_T(v)
#endif
struct FmWindow { int bx0, by0, bwp, bh, pitch, h, w, hs; };    // bwp / pitch: staged columns (pairs: bw - 2) and row pitch, in pairs / dwords
template <int NT> struct FmPixel {
    bool inimg;
    unsigned pix;           // element offset of the pixel inside an image plane
    float f[16];
    bool valid[NT];
    float alpha[NT], beta[NT];
    int lbase[NT];          // pair index of the 4x4 window origin of flow t inside the staged window
};

template <int K>
__device__ __forceinline__ void fm_wait_windows(int younger_groups) {
    switch (younger_groups) {
    case 0:  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K < 63 ? K : 63) : "memory"); break;
    case 2:  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K < 63 ? 2 * K : 63) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * K < 63 ? 3 * K : 63) : "memory"); break;
    }
}

// Channel loop of one workgroup on a window staged as pairs, written like fi_run_channels_lean (filterinterp_lds.hip): ring
// geometry a compile-time function of the window class and one constant s_waitcnt in the steady state, running plane pointers,
// tap reads as asm (8-byte pairs, see the head of the file) at immediate offsets from one address per evaluation, with one
// lgkmcnt wait per half evaluation and the next half's rows in flight under the current one's arithmetic, range-checked buffer
// stores.  An evaluation = one pixel under one flow: 2 x NT per thread and channel, each 8 LDS reads, 8 packed multiply-adds
// and the 4 operations of the blend.
// Window class <S, KR>: row pitch = 32 S pairs = S segments of 64 dwords, at most 8 KR rows.  A staging instruction writes one
// segment (64 consecutive dwords: what an LDS-DMA writes); wave v stages rows v, v + 8, ... whole: its KR x S instructions
// share S per-lane column offsets (the lane's column inside segment s, clamped to the frame; out of range for pad pairs) and
// take the row from the scalar offset operand -- S registers of addressing instead of one per staged dword (K = S KR = 9 ...
// 18), which is what lets a window of 96 x 48 pairs live beside 2 x 3 evaluations' state in 128 registers.  Rows past the
// window are staged through a descriptor of zero records: no memory traffic, and no address is formed from their offsets.
template <int S, int KR, int NT>
__device__ __forceinline__ void fm_run_channels(const float* __restrict__ img, const FmPtrs& ptr, int64_t boff, int64_t cs,
                                                int c_begin, int c_end, int tid, const FmWindow& win,
                                                const FmPixel<NT> (&px)[FM_PX], float* __restrict__ ring) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr int K = S * KR;                               // staging instructions per wave and channel
    constexpr int NP = K * FM_THREADS;                      // dwords per ring slot
    constexpr int R = (FM_RING_FLOATS / NP) < FM_RMAX ? (FM_RING_FLOATS / NP) : FM_RMAX;
    constexpr int D = R - 1;
    constexpr int NE = FM_PX * NT;                           // evaluations per thread and channel, e = t * FM_PX + p
    constexpr int PITCH8 = 256 * S;                          // row pitch in bytes
    static_assert(D >= 1 && D <= 4 && (D - 1) * K <= 63, "ring geometry");
    static_assert(3 * PITCH8 + 8 < 65536, "tap rows at immediate offsets");
    if (c_begin >= c_end) return;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // lane's dword inside segment s: half (lane & 1) of pair column 32 s + lane / 2
    unsigned voff[S];
#pragma unroll
    for (int s_ = 0; s_ < S; ++s_) {
        const int colp = 32 * s_ + (lane >> 1);
        voff[s_] = colp < win.bwp ? 4u * (unsigned)clampi(win.bx0 + colp + 2 * (lane & 1), 0, win.w - 1) : 0x80000000u;
    }
    const int plane_bytes = 4 * ((win.h - 1) * win.hs + win.w);
    const unsigned ring_lds = (unsigned)(uintptr_t)(fm_lptr_t)ring;
    unsigned lb[NE], pix4[FM_PX];
    float W[NE][4];                                          // blend4's weights of an evaluation: (1-a)(1-b), a(1-b), (1-a)b, ab
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < FM_PX; ++p) {
            const int e = t * FM_PX + p;
            lb[e] = ring_lds + 8u * (unsigned)px[p].lbase[t];                   // (an invalid evaluation's reads land anywhere: discarded)
            pix4[p] = 4u * px[p].pix;
            const float al = px[p].alpha[t], be = px[p].beta[t];
            W[e][0] = (1.0f - al) * (1.0f - be); W[e][1] = al * (1.0f - be); W[e][2] = (1.0f - al) * be; W[e][3] = al * be;
        }
    // filter taps as (left quadrant, right quadrant) pairs: rows 0-1 feed the top sums, rows 2-3 the bottom ones
    v2f F[FM_PX][8];
#pragma unroll
    for (int p = 0; p < FM_PX; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            F[p][2 * r] = v2f{px[p].f[4 * r], px[p].f[4 * r + 2]};
            F[p][2 * r + 1] = v2f{px[p].f[4 * r + 1], px[p].f[4 * r + 3]};
        }
    const int last = c_end - 1;
    const float* pdma = img + (int64_t)c_begin * cs;
    int64_t oofs = boff + (int64_t)c_begin * cs;            // element offset of the output plane inside every output tensor
    auto issue = [&](int slot) {
        float* l = ring + slot * NP + wave * (64 * S);
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int row = wave + 8 * k;                                       // (scalar)
            const bool live = row < win.bh;
            // (a row past the window: zero records -- every offset out of range)
            const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)pdma, 0, live ? plane_bytes : 0, 0x00020000);
            const int soff = live ? 4 * clampi(win.by0 + row, 0, win.h - 1) * win.hs : 0;
#pragma unroll
            for (int s_ = 0; s_ < S; ++s_)
                if (!(FM_ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (fm_lptr_t)(l + (8 * k * S + s_) * 64), 4, voff[s_], soff, 0, 0);
        }
        pdma += cs;
    };
#define FM_READ_ROW(d0, d1, addr, r) do { \
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d0) : "v"(addr), "n"((r) * PITCH8)); \
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d1) : "v"(addr), "n"((r) * PITCH8 + 8)); } while (0)
    auto compute = [&](int slot) {
        const unsigned so = (unsigned)(slot * (NP * 4));
        // Tap reads ping-pong between two register sets.  Three flows (development builds): half an evaluation per set -- while the
        // top sums of evaluation e are formed from rows 0-1 (set 0), rows 2-3 (set 1) are in flight, and so on; whole
        // evaluations in flight do not fit beside 2 x 3 evaluations' state at 128 registers.  Two flows could hold a whole
        // evaluation per set (FM_DEEP: 16 LDS reads in flight, what the 4-bit lgkmcnt can count) -- measured slower, off.
        constexpr bool DEEP = NT < 3 && FM_DEEP;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I4 = std::integral_constant<int, 4>;
        auto finish = [&](auto E, const v2f& top, const v2f& bot) {
            constexpr int e = decltype(E)::value, t = e / FM_PX, p = e % FM_PX;
            float val = W[e][0] * top.x;                    // (blend4, its weights formed above)
            val = fmaf(W[e][1], top.y, val);
            val = fmaf(W[e][2], bot.x, val);
            val = fmaf(W[e][3], bot.y, val);
            const auto oplane = __builtin_amdgcn_make_buffer_rsrc((void*)(ptr.out[t] + oofs), 0, plane_bytes, 0x00020000);
            // an invalid evaluation's store is dropped by the range check.  (The select is formed here, from a validity mask
            // the compiler keeps in scalar registers: loop-invariant offsets per evaluation instead of per pixel cost registers.)
            unsigned po = pix4[p];
            asm volatile("" : "+v"(po));
            if (!(FM_ABL & 1) || val == 123456.789f)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), oplane, px[p].valid[t] ? po : 0x80000000u, 0, 0);
        };
        if constexpr (DEEP) {
            v2f q[2][8];
            auto reads_all = [&](auto E) {
                constexpr int e = decltype(E)::value;
                v2f (&d)[8] = q[e & 1];
                const unsigned a = lb[e] + so;
                if (FM_ABL & 4) { for (int i = 0; i < 8; ++i) d[i] = v2f{__uint_as_float(a), 1.0f}; return; }
                FM_READ_ROW(d[0], d[1], a, 0); FM_READ_ROW(d[2], d[3], a, 1);
                FM_READ_ROW(d[4], d[5], a, 2); FM_READ_ROW(d[6], d[7], a, 3);
            };
            reads_all(I0{}); reads_all(I1{});
            static_for<0, NE>([&](auto E) {
                constexpr int e = decltype(E)::value, p = e % FM_PX;
                v2f (&d)[8] = q[e & 1];
                // (LDS reads return in order: with the next evaluation's eight behind them, this one's are back at lgkmcnt(8))
                asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                                                        "+v"(d[6]), "+v"(d[7]) : "n"(e + 1 < NE ? 8 : 0));
                v2f top = d[0] * F[p][0];
                top = __builtin_elementwise_fma(d[1], F[p][1], top);
                top = __builtin_elementwise_fma(d[2], F[p][2], top);
                top = __builtin_elementwise_fma(d[3], F[p][3], top);
                v2f bot = d[4] * F[p][4];
                bot = __builtin_elementwise_fma(d[5], F[p][5], bot);
                bot = __builtin_elementwise_fma(d[6], F[p][6], bot);
                bot = __builtin_elementwise_fma(d[7], F[p][7], bot);
                if constexpr (e + 2 < NE) reads_all(std::integral_constant<int, e + 2>{});
                finish(E, top, bot);
            });
        } else {
        v2f q[2][4];
        unsigned adr[2];
        auto reads = [&](auto E, auto H) {                   // rows 2h, 2h + 1 of evaluation e into set h
            constexpr int e = decltype(E)::value, h = decltype(H)::value;
            v2f (&d)[4] = q[h];
            if constexpr (h == 0) adr[e & 1] = lb[e] + so;
            const unsigned a = adr[e & 1];
            if (FM_ABL & 4) { d[0] = d[1] = d[2] = d[3] = v2f{__uint_as_float(a), 1.0f}; return; }
            FM_READ_ROW(d[0], d[1], a, 2 * h);
            FM_READ_ROW(d[2], d[3], a, 2 * h + 1);
        };
        auto landed = [&](v2f (&d)[4], auto LATER) {         // all but the `later` youngest LDS reads are back
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "n"(decltype(LATER)::value));
        };
        reads(I0{}, I0{}); reads(I0{}, I1{});
        static_for<0, NE>([&](auto E) {
            constexpr int e = decltype(E)::value, p = e % FM_PX;
            // d[2r] = columns (0, 2), d[2r + 1] = columns (1, 3) of row r: the sums of fi4_pixel, two quadrants per instruction
            landed(q[0], I4{});
            v2f top = q[0][0] * F[p][0];
            top = __builtin_elementwise_fma(q[0][1], F[p][1], top);
            top = __builtin_elementwise_fma(q[0][2], F[p][2], top);
            top = __builtin_elementwise_fma(q[0][3], F[p][3], top);
            if constexpr (e + 1 < NE) { reads(std::integral_constant<int, e + 1>{}, I0{}); landed(q[1], I4{}); }
            else landed(q[1], I0{});
            v2f bot = q[1][0] * F[p][4];
            bot = __builtin_elementwise_fma(q[1][1], F[p][5], bot);
            bot = __builtin_elementwise_fma(q[1][2], F[p][6], bot);
            bot = __builtin_elementwise_fma(q[1][3], F[p][7], bot);
            if constexpr (e + 1 < NE) reads(std::integral_constant<int, e + 1>{}, I1{});
            finish(E, top, bot);
        });
        }
        oofs += cs;
    };
#undef FM_READ_ROW
    // The steady-state wait.  vmcnt counts the staging loads AND the NE result stores of every step, in issue order (one counter
    // on gfx9).  Behind the loads of window c + 1 the stream holds, at the end of step c: the stores of the step that issued
    // them, then D - 1 steps of K loads + NE stores.  Rounds 3-4 waited for vmcnt <= (D - 1) K -- with the stores in the counter
    // that asks for part of window c + 2 as well, and with two ring slots (D = 1) for every store of the step itself.  The
    // prologue puts NE stores whose offsets are out of range behind each of its windows, so that the first steps see the same
    // stream as the later ones.  (In-kernel stamps had a quarter of a step in this wait.)
#ifndef FM_COUNT_STORES
#define FM_COUNT_STORES 1
#endif
    constexpr int NWAIT = (FM_COUNT_STORES && !(FM_ABL & 1)) ? (D - 1) * (K + NE) + NE : (D - 1) * K;
    static_assert(NWAIT <= 63, "vmcnt is six bits");
    const int n0 = min(D, c_end - c_begin);
    {
        const auto nowhere = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, 0, 0x00020000);      // (zero records: every offset out of range)
        for (int j = 0; j < n0; ++j) {
            issue(j);
            if (FM_COUNT_STORES) {
#pragma unroll
                for (int e = 0; e < NE; ++e) __builtin_amdgcn_raw_buffer_store_b32(0u, nowhere, 0x80000000u, 0, 0);
            }
        }
    }
    fm_wait_windows<K>(n0 - 1);                                 // the first window has landed ...
    __builtin_amdgcn_s_barrier();                               // ... in every wave
    int c = c_begin, slot = 0;
#ifdef FM_STAMPS
    unsigned long long acc_i = 0, acc_c = 0, acc_w = 0, acc_b = 0, acc_n = 0;
#endif
    for (; c + D <= last; ++c) {                                // steady state: window c + D exists
        FM_T(t0);
        issue(slot == 0 ? R - 1 : slot - 1);                    // into the slot every wave finished reading before the last barrier
        FM_T(t1);
        compute(slot);
        FM_T(t2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");            // window c + 1 has landed
        FM_T(t3);
        __builtin_amdgcn_s_barrier();
#ifdef FM_STAMPS
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        acc_i += t1 - t0; acc_c += t2 - t1; acc_w += t3 - t2; acc_b += t4 - t3; acc_n += 1;
#endif
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
#ifdef FM_STAMPS
    if (lane == 0) {
        atomicAdd(&g_fm_stamps[0], acc_i); atomicAdd(&g_fm_stamps[1], acc_c); atomicAdd(&g_fm_stamps[2], acc_w);
        atomicAdd(&g_fm_stamps[3], acc_b); atomicAdd(&g_fm_stamps[4], acc_n);
    }
#endif
    for (; c <= last; ++c) {                                    // the last D channels: nothing left to stage
        compute(slot);
        if (c < last) {
            // behind the loads of window c + 1: m windows (m = last - c - 1 <= D - 2) and the stores of D steps -- provided
            // window c + 1 was staged by a step of the loop above (a channel range shorter than the ring waits for everything)
            const int m = last - c - 1;
            if (!FM_COUNT_STORES || (FM_ABL & 1) || c - c_begin < D - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (m <= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * NE < 63 ? D * NE : 63) : "memory");
            else if (m == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K + D * NE < 63 ? K + D * NE : 63) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K + D * NE < 63 ? 2 * K + D * NE : 63) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    // copy-through of the invalid pixels (:2814-2818), outside the pipelined loop
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < FM_PX; ++p)
            if (px[p].inimg && !px[p].valid[t])
                for (int cc = c_begin; cc < c_end; ++cc)
                    ptr.out[t][boff + (int64_t)cc * cs + px[p].pix] = img[(int64_t)cc * cs + px[p].pix];
}

// The same loop on a PLAIN window (one dword per element, tap rows as two ds_read2_b32: rounds 2-3) -- for tiles whose window
// does not fit the ring as pairs (rough flow fields: a window of 98 x 50 elements is 12.5 dwords per thread plain, 25 as
// pairs).  Written like fi_run_channels_lean (filterinterp_lds.hip): ring geometry a compile-time function of K and one constant s_waitcnt in the steady state, running
// plane pointers, M0 formed on the scalar unit, tap reads as asm (columns (0, 2) / (1, 3) of a row, so that one packed
// multiply-add advances the left and the right quadrant sum) with one lgkmcnt wait per evaluation and the next evaluation's
// first rows in flight under the current one's arithmetic, range-checked buffer stores.  An evaluation = one pixel under
// one flow: 2 x NT per thread and channel.
template <int K, int NT>
__device__ __forceinline__ void fm_run_channels_plain(const float* __restrict__ img, const FmPtrs& ptr, int64_t boff, int64_t cs,
                                                int c_begin, int c_end, int tid, const FmWindow& win,
                                                const FmPixel<NT> (&px)[FM_PX], float* __restrict__ ring) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr int NP = K * FM_THREADS;
    constexpr int R = (FM_RING_FLOATS / NP) < FM_RMAX ? (FM_RING_FLOATS / NP) : FM_RMAX;
    constexpr int D = R - 1;
    constexpr int NE = FM_PX * NT;                           // evaluations per thread and channel, e = t * FM_PX + p
    static_assert(D >= 1 && D <= 4 && (D - 1) * K <= 63, "ring geometry");
    if (c_begin >= c_end) return;
    // staged element e = tid + k * FM_THREADS, row pitch a multiple of the 32 LDS banks, borders replicated while
    // staging, pad elements out of the buffer's range (they cost no memory traffic): filterinterp_lds.hip
    const float inv_pitch32 = 1.0f / (float)win.pitch;
    unsigned goff[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int e = tid + k * FM_THREADS;
        const int r = fi_row_of(e, inv_pitch32);
        const int col = e - r * win.pitch;
        const unsigned off = 4u * (unsigned)(clampi(win.by0 + r, 0, win.h - 1) * win.hs + clampi(win.bx0 + col, 0, win.w - 1));
        goff[k] = (col < win.bwp && r < win.bh) ? off : 0x80000000u;    // (bwp = the window width here)
    }
    const int plane_bytes = 4 * ((win.h - 1) * win.hs + win.w);
    const int wave_first = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
    const unsigned ring_lds = (unsigned)(uintptr_t)(fm_lptr_t)ring;
    const unsigned pitch4 = 4u * (unsigned)win.pitch;
    unsigned lb[NE], pix4[FM_PX];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < FM_PX; ++p) {
            lb[t * FM_PX + p] = ring_lds + 4u * (unsigned)px[p].lbase[t];       // (an invalid evaluation's reads land anywhere: discarded)
            pix4[p] = 4u * px[p].pix;
        }
    // filter taps as (left quadrant, right quadrant) pairs: rows 0-1 feed the top sums, rows 2-3 the bottom ones (two flows;
    // with three, 16 more aligned register pairs are more than the allocator places without spilling inside the loop)
    constexpr bool PACKED = NT < 3;
    v2f F[FM_PX][8];
    if constexpr (PACKED) {
#pragma unroll
        for (int p = 0; p < FM_PX; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                F[p][2 * r] = v2f{px[p].f[4 * r], px[p].f[4 * r + 2]};
                F[p][2 * r + 1] = v2f{px[p].f[4 * r + 1], px[p].f[4 * r + 3]};
            }
    }
    const int last = c_end - 1;
    const float* pdma = img + (int64_t)c_begin * cs;
    int64_t oofs = boff + (int64_t)c_begin * cs;            // element offset of the output plane inside every output tensor
    auto issue = [&](int slot) {
        const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)pdma, 0, plane_bytes, 0x00020000);
        float* l = ring + slot * NP + wave_first;
#pragma unroll
        for (int k = 0; k < K; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (fm_lptr_t)(l + k * FM_THREADS), 4, goff[k], 0, 0, 0);
        pdma += cs;
    };
#define FM_READ2(dst, addr, o0, o1) asm volatile("ds_read2_b32 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(dst) : "v"(addr))
    auto compute = [&](int slot) {
        const unsigned so = (unsigned)(slot * (NP * 4));
        // (a second evaluation in flight costs 16 registers: with three flows the allocator keeps it without spilling inside
        //  the loop for one ring geometry only -- measured 2.91 against 2.97 ms per C=196 launch where it does)
        constexpr bool OVERLAP = NT < 3 ? K <= 12 : K == 5;
        v2f q[OVERLAP ? 2 : 1][8];
        auto reads = [&](auto E, auto H) {                   // rows 2h, 2h + 1 of evaluation e
            constexpr int e = decltype(E)::value, h = decltype(H)::value;
            v2f (&d)[8] = q[OVERLAP ? (e & 1) : 0];
            unsigned a = lb[e] + so + (h ? 2u * pitch4 : 0u);
            FM_READ2(d[4 * h], a, 0, 2); FM_READ2(d[4 * h + 1], a, 1, 3);
            a += pitch4;
            FM_READ2(d[4 * h + 2], a, 0, 2); FM_READ2(d[4 * h + 3], a, 1, 3);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        reads(I0{}, I0{}); reads(I0{}, I1{});
        if constexpr (OVERLAP) reads(I1{}, I0{});
        static_for<0, NE>([&](auto E) {
            constexpr int e = decltype(E)::value, t = e / FM_PX, p = e % FM_PX;
            v2f (&d)[8] = q[OVERLAP ? (e & 1) : 0];
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]),
                                                    "+v"(d[6]), "+v"(d[7]) : "n"(OVERLAP && e + 1 < NE ? 4 : 0));
            v2f top, bot;
            if constexpr (PACKED) {
                top = d[0] * F[p][0];                       // (same order per quadrant sum as fi4_pixel)
                top = __builtin_elementwise_fma(d[1], F[p][1], top);
                top = __builtin_elementwise_fma(d[2], F[p][2], top);
                top = __builtin_elementwise_fma(d[3], F[p][3], top);
                bot = d[4] * F[p][4];
                bot = __builtin_elementwise_fma(d[5], F[p][5], bot);
                bot = __builtin_elementwise_fma(d[6], F[p][6], bot);
                bot = __builtin_elementwise_fma(d[7], F[p][7], bot);
            } else {
                // d[2r] = columns (0, 2), d[2r + 1] = columns (1, 3) of row r; the sums of fi4_pixel, one float at a time
                const float (&f)[16] = px[p].f;
                float TL = d[0].x * f[0];  TL = fmaf(d[1].x, f[1], TL);  TL = fmaf(d[2].x, f[4], TL);   TL = fmaf(d[3].x, f[5], TL);
                float TR = d[0].y * f[2];  TR = fmaf(d[1].y, f[3], TR);  TR = fmaf(d[2].y, f[6], TR);   TR = fmaf(d[3].y, f[7], TR);
                float BL = d[4].x * f[8];  BL = fmaf(d[5].x, f[9], BL);  BL = fmaf(d[6].x, f[12], BL);  BL = fmaf(d[7].x, f[13], BL);
                float BR = d[4].y * f[10]; BR = fmaf(d[5].y, f[11], BR); BR = fmaf(d[6].y, f[14], BR);  BR = fmaf(d[7].y, f[15], BR);
                top = v2f{TL, TR}; bot = v2f{BL, BR};
            }
            // (the four blend weights are formed here each time: kept across the loop for 2 x NT evaluations they do not fit)
            float al = px[p].alpha[t], be = px[p].beta[t];
            asm volatile("" : "+v"(al), "+v"(be));
            const float val = blend4(al, be, top.x, top.y, bot.x, bot.y);
            const auto oplane = __builtin_amdgcn_make_buffer_rsrc((void*)(ptr.out[t] + oofs), 0, plane_bytes, 0x00020000);
            unsigned po = pix4[p];                          // (an invalid evaluation's store is dropped by the range check)
            asm volatile("" : "+v"(po));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), oplane, px[p].valid[t] ? po : 0x80000000u, 0, 0);
            if constexpr (OVERLAP) {
                if constexpr (e + 1 < NE) reads(std::integral_constant<int, e + 1>{}, I1{});
                if constexpr (e + 2 < NE) reads(std::integral_constant<int, e + 2>{}, I0{});
            } else if constexpr (e + 1 < NE) {
                reads(std::integral_constant<int, e + 1>{}, I0{}); reads(std::integral_constant<int, e + 1>{}, I1{});
            }
        });
        oofs += cs;
    };
#undef FM_READ2
    // (the steady-state wait counts the result stores in: see fm_run_channels)
    constexpr int NWAIT = FM_COUNT_STORES ? (D - 1) * (K + NE) + NE : (D - 1) * K;
    static_assert(NWAIT <= 63, "vmcnt is six bits");
    const int n0 = min(D, c_end - c_begin);
    {
        const auto nowhere = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, 0, 0x00020000);
        for (int j = 0; j < n0; ++j) {
            issue(j);
            if (FM_COUNT_STORES) {
#pragma unroll
                for (int e = 0; e < NE; ++e) __builtin_amdgcn_raw_buffer_store_b32(0u, nowhere, 0x80000000u, 0, 0);
            }
        }
    }
    fm_wait_windows<K>(n0 - 1);                                 // the first window has landed ...
    __builtin_amdgcn_s_barrier();                               // ... in every wave
    int c = c_begin, slot = 0;
    for (; c + D <= last; ++c) {                                // steady state: window c + D exists
        issue(slot == 0 ? R - 1 : slot - 1);                    // into the slot every wave finished reading before the last barrier
        compute(slot);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");            // window c + 1 has landed
        __builtin_amdgcn_s_barrier();
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    for (; c <= last; ++c) {                                    // the last D channels: nothing left to stage
        compute(slot);
        if (c < last) {
            // behind the loads of window c + 1: m windows (m = last - c - 1 <= D - 2) and the stores of D steps -- provided
            // window c + 1 was staged by a step of the loop above (a channel range shorter than the ring waits for everything)
            const int m = last - c - 1;
            if (!FM_COUNT_STORES || (FM_ABL & 1) || c - c_begin < D - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (m <= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * NE < 63 ? D * NE : 63) : "memory");
            else if (m == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K + D * NE < 63 ? K + D * NE : 63) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * K + D * NE < 63 ? 2 * K + D * NE : 63) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        slot = (slot + 1 == R) ? 0 : slot + 1;
    }
    // copy-through of the invalid pixels (:2814-2818), outside the pipelined loop
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < FM_PX; ++p)
            if (px[p].inimg && !px[p].valid[t])
                for (int cc = c_begin; cc < c_end; ++cc)
                    ptr.out[t][boff + (int64_t)cc * cs + px[p].pix] = img[(int64_t)cc * cs + px[p].pix];
}

template <int NT>
__global__ __launch_bounds__(FM_THREADS, 4) void fi_forward_ori_multi(
    const float* __restrict__ in1, FmPtrs ptr, const float* __restrict__ in3, int channel, int h, int w,
    vfi_strides s1, vfi_strides s2, vfi_strides s3, int tiles_x, int tiles_y, int ntiles, int ch_per_group, int kpair) {
    __shared__ float lds[FM_HDR + FM_RING_FLOATS];
    int* box = reinterpret_cast<int*>(lds);

    // four horizontally consecutive tiles per XCD (filterinterp_lds.hip)
    const int bid = blockIdx.x;
    const int xs = bid % FM_XCDS, kk = bid / FM_XCDS;
    const int tile = ((kk / 4) * FM_XCDS + xs) * 4 + (kk % 4);
    if (tile >= ntiles) return;
    const int b = tile / (tiles_x * tiles_y);
    const int trem = tile - b * (tiles_x * tiles_y);
    const int tyi = trem / tiles_x, txi = trem - tyi * tiles_x;
    const int c_begin = blockIdx.y * ch_per_group;
    const int c_end = min(channel, c_begin + ch_per_group);

    const int tid = threadIdx.x;
    const int x = txi * FM_TW + (tid & (FM_TW - 1));
    const int y0 = tyi * FM_TH + (tid >> 6);

    const int flow_bytes = ((int)s2.c + (h - 1) * (int)s2.h + w) * 4;           // (the host checked that these fit 31 bits)
    const int filt_bytes = (15 * (int)s3.c + (h - 1) * (int)s3.h + w) * 4;
    FmPixel<NT> px[FM_PX];
    int L[FM_PX][NT], T[FM_PX][NT];
    float fxv[FM_PX][NT], fyv[FM_PX][NT];
    int bx_lo = INT_MAX, by_lo = INT_MAX, bx_hi = INT_MIN, by_hi = INT_MIN;
#pragma unroll
    for (int p = 0; p < FM_PX; ++p) {
        const int y = y0 + p * FM_PASS_ROWS;
        px[p].inimg = x < w && y < h;
        px[p].pix = (unsigned)(y * (int)s1.h + x);
#pragma unroll
        for (int k = 0; k < 16; ++k) px[p].f[k] = 0.0f;
#pragma unroll
        for (int t = 0; t < NT; ++t) fxv[p][t] = fyv[p][t] = 0.0f;
        if (px[p].inimg) {
            // (buffer loads: a uniform descriptor per tensor, a 32-bit byte offset per pixel, the plane's offset scalar)
            const int fo = (y * (int)s2.h + x) * 4, ko = (y * (int)s3.h + x) * 4;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const auto fr = __builtin_amdgcn_make_buffer_rsrc((void*)(ptr.flow[t] + (int64_t)b * s2.b), 0, flow_bytes, 0x00020000);
                fxv[p][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fr, fo, 0, 0));
                fyv[p][t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fr, fo, (int)s2.c * 4, 0));
            }
            const auto kr = __builtin_amdgcn_make_buffer_rsrc((void*)(in3 + (int64_t)b * s3.b), 0, filt_bytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < 16; ++k) px[p].f[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(kr, ko, k * (int)s3.c * 4, 0));
        }
    }
#pragma unroll
    for (int p = 0; p < FM_PX; ++p) {
        const int y = y0 + p * FM_PASS_ROWS;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float fx = fxv[p][t], fy = fyv[p][t];
            const float x2 = (float)x + fx;
            const float y2 = (float)y + fy;
            px[p].valid[t] = px[p].inimg && fi_valid(fx, fy, x2, y2, w, h);
            const int ix = px[p].valid[t] ? (int)x2 : 0, iy = px[p].valid[t] ? (int)y2 : 0;
            L[p][t] = ix - 1;                               // ix + 1 - fs/2, fs == 4
            T[p][t] = iy - 1;
            px[p].alpha[t] = x2 - (float)ix;
            px[p].beta[t] = y2 - (float)iy;
            if (px[p].valid[t]) {
                bx_lo = min(bx_lo, L[p][t]); by_lo = min(by_lo, T[p][t]);
                bx_hi = max(bx_hi, L[p][t] + 3); by_hi = max(by_hi, T[p][t] + 3);
            }
        }
    }

    // ---- bounding box of every tap of the tile, all flows
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
    // the slot holds pairs (column c, column c + 2): bw - 2 of them per row, pitch a multiple of 32 pairs; a window too large
    // for that is staged plain (pitch a multiple of 32 dwords)
    const int bwp = any_valid ? bw - 2 : 0;
    const int segs = max(3, (bwp + 31) >> 5);                // S: segments of 32 pairs per row (a narrow window takes the class of 3)
    const int rows8 = max(3, (bh + 7) >> 3);                 // KR: rows per staging wave
    const bool paired = segs <= 4 && segs * rows8 <= kpair;  // (kpair <= FM_KPAIR)
    const int pitch = paired ? 32 * segs : (bw + 31) & ~31;
    const int n = paired ? 0 : pitch * bh;                   // dwords of a plain window
#pragma unroll
    for (int p = 0; p < FM_PX; ++p)
#pragma unroll
        for (int t = 0; t < NT; ++t) px[p].lbase[t] = (T[p][t] - by0) * pitch + (L[p][t] - bx0);

    const int64_t boff = (int64_t)b * s1.b;
    const float* img = in1 + boff;
    const int kmax = (n + FM_THREADS - 1) / FM_THREADS;
    if (kmax > FM_KTOP) {
        // window too large for LDS: gather from global memory (workgroup-uniform branch)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int p = 0; p < FM_PX; ++p) {
                float* dst = ptr.out[t] + boff;
                if (px[p].valid[t]) {
                    fi4_channels_direct(img, dst + px[p].pix, c_begin, c_end, s1.c, (int)s1.h, h, w, L[p][t], T[p][t], px[p].f,
                                        px[p].alpha[t], px[p].beta[t]);
                } else if (px[p].inimg) {
                    for (int c = c_begin; c < c_end; ++c) dst[(int64_t)c * s1.c + px[p].pix] = img[(int64_t)c * s1.c + px[p].pix];
                }
            }
        return;
    }

    const FmWindow win{bx0, by0, paired ? bwp : bw, bh, pitch, h, w, (int)s1.h};
    float* ring = lds + FM_HDR;
#define FM_RUN(S, KR) fm_run_channels<S, KR, NT>(img, ptr, boff, s1.c, c_begin, c_end, tid, win, px, ring)
#define FM_RUN_PLAIN(K) fm_run_channels_plain<K, NT>(img, ptr, boff, s1.c, c_begin, c_end, tid, win, px, ring)
    if (paired) {
        if (segs == 3) {
            if (rows8 == 3) FM_RUN(3, 3);
            else if (rows8 == 4) FM_RUN(3, 4);
            else if (rows8 == 5) FM_RUN(3, 5);
            else FM_RUN(3, 6);
        } else {
            if (rows8 == 3) FM_RUN(4, 3);
            else FM_RUN(4, 4);
        }
    } else {
        if (kmax <= 8) FM_RUN_PLAIN(8);
        else if (kmax <= 10) FM_RUN_PLAIN(10);
        else if (kmax <= 12) FM_RUN_PLAIN(12);
        else FM_RUN_PLAIN(15);
    }
#undef FM_RUN_PLAIN
#undef FM_RUN
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_filterinterp_forward_ori(const float* input1, const float* input2, const float* input3,
                                             float* output, int batch, int channel, int h, int w,
                                             int filter_channels, vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                             vfi_stream_t stream);

// largest staged dwords per thread a window may need as pairs (development builds: 0 = every window plain)
VFI_KNOB(int, g_fm_kpair, FM_KPAIR);
// flows per launch, at most (2 or 3)
VFI_KNOB(int, g_fm_group, FM_GROUP);
#ifdef VFI_DEV
extern "C" void vfi_dev_multi(int kpair, int group) {
    g_fm_kpair = kpair < FM_KPAIR ? kpair : FM_KPAIR;
    if (group == 2 || group == 3) g_fm_group = group;
}
#endif
#ifdef FM_STAMPS
// reads the accumulators and clears them (synchronises)
extern "C" int vfi_dev_multi_stamps(unsigned long long* host8) {
    if (hipDeviceSynchronize() != hipSuccess) return VFI_ERR_LAUNCH;
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_fm_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return VFI_ERR_LAUNCH;
    const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fm_stamps), zero, sizeof(zero)) == hipSuccess ? VFI_OK : VFI_ERR_LAUNCH;
}
#endif

extern "C" int vfi_filterinterp_forward_ori_multi(const float* input1, const float* const* flows, const float* input3,
                                                   float* const* outputs, int nflows, int batch, int channel, int h, int w,
                                                   int filter_channels, vfi_strides s1, vfi_strides s2, vfi_strides s3,
                                                   vfi_stream_t stream) {
    if (nflows <= 0 || !flows || !outputs || batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_channels <= 0)
        return VFI_ERR_SHAPE;
    if (!input1 || !input3) return VFI_ERR_SHAPE;
    for (int t = 0; t < nflows; ++t)
        if (!flows[t] || !outputs[t]) return VFI_ERR_SHAPE;
    // more flows than one launch takes: groups of g_fm_group, the rest as a pair or alone
    if (nflows > g_fm_group) {
        for (int t0 = 0; t0 < nflows;) {
            const int n = (g_fm_group == 3 && nflows - t0 == 4) ? 2 : (nflows - t0 < g_fm_group ? nflows - t0 : g_fm_group);
            const int err = vfi_filterinterp_forward_ori_multi(input1, flows + t0, input3, outputs + t0, n, batch, channel, h, w,
                                                               filter_channels, s1, s2, s3, stream);
            if (err != VFI_OK) return err;
            t0 += n;
        }
        return VFI_OK;
    }
    // the shared-window kernel: fs == 4, 2 or 3 flows, in-plane byte offsets within 32 bits; anything else is the
    // single-flow entry point once per flow (same results)
    const bool staged = filter_channels == 16 && nflows >= 2 && nflows <= g_fm_group && (int64_t)h * s1.h * 4 <= INT_MAX &&
                        s2.c >= 0 && s3.c >= 0 && (s2.c + (int64_t)h * s2.h) * 4 <= INT_MAX && (15 * s3.c + (int64_t)h * s3.h) * 4 <= INT_MAX;
    if (!staged) {
        for (int t = 0; t < nflows; ++t) {
            const int err = vfi_filterinterp_forward_ori(input1, flows[t], input3, outputs[t], batch, channel, h, w,
                                                         filter_channels, s1, s2, s3, stream);
            if (err != VFI_OK) return err;
        }
        return VFI_OK;
    }
    const int tiles_x = (w + FM_TW - 1) / FM_TW, tiles_y = (h + FM_TH - 1) / FM_TH;
    const int64_t nt = (int64_t)tiles_x * tiles_y * batch;
    if (nt > (1 << 28)) return VFI_ERR_SHAPE;
    const int ntiles = (int)nt;
    const int per_xcd = (((ntiles + FM_XCDS - 1) / FM_XCDS) + 3) & ~3;          // whole groups of four tiles
    // (one prologue for nflows outputs per channel)
    const int best_groups = fi_channel_groups(ntiles, channel, 4.3 * (1.0 + 0.3 * (nflows - 1)) / nflows);
    const int ch_per_group = (channel + best_groups - 1) / best_groups;
    const int groups = (channel + ch_per_group - 1) / ch_per_group;
    FmPtrs ptr;
    for (int t = 0; t < FM_MAXT; ++t) { ptr.flow[t] = flows[t < nflows ? t : 0]; ptr.out[t] = outputs[t < nflows ? t : 0]; }
    const dim3 grid((unsigned)(per_xcd * FM_XCDS), (unsigned)groups, 1), block(FM_THREADS, 1, 1);
    hipStream_t st = (hipStream_t)stream;
    switch (nflows) {
#ifdef VFI_DEV
    case 3: hipLaunchKernelGGL(fi_forward_ori_multi<3>, grid, block, 0, st, input1, ptr, input3, channel, h, w, s1, s2, s3, tiles_x, tiles_y, ntiles, ch_per_group, g_fm_kpair); break;
#endif
    default: hipLaunchKernelGGL(fi_forward_ori_multi<2>, grid, block, 0, st, input1, ptr, input3, channel, h, w, s1, s2, s3, tiles_x, tiles_y, ntiles, ch_per_group, g_fm_kpair); break;
    }
    return launch_status();
}
