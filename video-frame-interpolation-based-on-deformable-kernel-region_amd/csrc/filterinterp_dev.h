// filterinterp_dev.h -- per-pixel device code shared by the direct and the
// LDS-staged FilterInterpolation (_ori, fs == 4) forward kernels.
#pragma once
#include "vfi_common.h"

namespace vfi {

// One pixel's 4x4 window, fs == 4: every quadrant is 2x2.  v = the 16 image
// taps (row major), f = the 16 filter taps.  Accumulation order inside each
// quadrant is rows outer, columns inner (filterinterpolation_cuda_kernel.cu:2749-2787);
// `acc += a*b` fused as nvcc -fmad=true fuses it.
__device__ __forceinline__ float fi4_pixel(const float (&v)[16], const float (&f)[16], float alpha, float beta) {
    float TL = v[0] * f[0];   TL = fmaf(v[1], f[1], TL);   TL = fmaf(v[4], f[4], TL);   TL = fmaf(v[5], f[5], TL);
    float TR = v[2] * f[2];   TR = fmaf(v[3], f[3], TR);   TR = fmaf(v[6], f[6], TR);   TR = fmaf(v[7], f[7], TR);
    float BL = v[8] * f[8];   BL = fmaf(v[9], f[9], BL);   BL = fmaf(v[12], f[12], BL); BL = fmaf(v[13], f[13], BL);
    float BR = v[10] * f[10]; BR = fmaf(v[11], f[11], BR); BR = fmaf(v[14], f[14], BR); BR = fmaf(v[15], f[15], BR);
    return blend4(alpha, beta, TL, TR, BL, BR);
}

// Row of staged element e (e < 2^15) in a window of row pitch p: floor(e / p) through a float reciprocal -- (e + 0.5) / p never
// comes within 0.5 / p of an integer, while the two roundings are off by less than 2e-3 / p (the quotient is at most 2^15 / p)
// -- instead of a 32-bit integer division (~40 instructions, once per staged element and tile).
__device__ __forceinline__ int fi_row_of(int e, float inv_pitch) {
    return (int)(((float)e + 0.5f) * inv_pitch);
}

// LDS row pitch of a staged window bw elements (dwords) wide: a multiple of the 32 banks, plus FI_PITCH_SKEW.
// The LDS serves the 4-byte tap reads 32 lanes per clock on 32 banks (measured with rocprofv3's SQ_LDS_BANK_CONFLICT on
// controlled flow fields, tools/lds_conflicts.sh).  fp32 windows: 32 neighbouring lanes read 32 different columns, so with a
// pitch that is a multiple of 32 a tap's bank depends on its column only and lanes that sit on different window rows (a step
// in the flow's vertical part) cannot collide -- 0 conflict cycles on such fields, 50 % with a skew of 8 or 16; what is left
// on a smooth field are the lanes a stretching flow pushes onto a 33rd column.  fp16 windows: two neighbouring lanes share
// a dword -- a broadcast while they sit on one row, a two-way conflict when the step falls between them (50 % conflict
// cycles on a field with a step per 20 lanes, 45 % on the smooth field).  There the 32 lanes use 16-17 banks, and a skew of
// 16 dwords puts the next row's copy of a dword on a bank none of them uses: conflict cycles on the smooth field 138 M ->
// 63 M per C=196 launch (-4.5 % time), on the quarter field 637 M -> 412 M (-20 %).
#ifndef FI_PITCH_SKEW
#define FI_PITCH_SKEW 0
#endif
__device__ __forceinline__ int fi_pitch_for(int bw) {
    return FI_PITCH_SKEW ? (((max(bw - FI_PITCH_SKEW, 0) + 31) & ~31) + FI_PITCH_SKEW) : ((bw + 31) & ~31);
}

// channel loop of one valid pixel gathering straight from global memory.  Row by row, so the
// register need is 4 taps + 4 sums (the compiler may still batch rows when it has registers
// to spare); the operation order per quadrant is that of fi4_pixel.
__device__ __forceinline__ void fi4_channels_direct(const float* __restrict__ img, float* __restrict__ dst,
                                                    int c0, int c1, int64_t cs, int hs, int h, int w,
                                                    int L, int T, const float (&f)[16], float alpha, float beta) {
    // unsigned 32-bit element offsets from a wave-uniform plane pointer: the loads take the
    // scalar-base + vector-offset form, one VGPR per address
    unsigned ro[4], co[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ro[k] = (unsigned)(clampi(T + k, 0, h - 1) * hs);
        co[k] = (unsigned)clampi(L + k, 0, w - 1);
    }
    for (int c = c0; c < c1; ++c) {
        const float* p = img + (int64_t)c * cs;
        float a0 = p[ro[0] + co[0]], a1 = p[ro[0] + co[1]], a2 = p[ro[0] + co[2]], a3 = p[ro[0] + co[3]];
        float TL = a0 * f[0];  TL = fmaf(a1, f[1], TL);
        float TR = a2 * f[2];  TR = fmaf(a3, f[3], TR);
        a0 = p[ro[1] + co[0]]; a1 = p[ro[1] + co[1]]; a2 = p[ro[1] + co[2]]; a3 = p[ro[1] + co[3]];
        TL = fmaf(a0, f[4], TL);  TL = fmaf(a1, f[5], TL);
        TR = fmaf(a2, f[6], TR);  TR = fmaf(a3, f[7], TR);
        a0 = p[ro[2] + co[0]]; a1 = p[ro[2] + co[1]]; a2 = p[ro[2] + co[2]]; a3 = p[ro[2] + co[3]];
        float BL = a0 * f[8];   BL = fmaf(a1, f[9], BL);
        float BR = a2 * f[10];  BR = fmaf(a3, f[11], BR);
        a0 = p[ro[3] + co[0]]; a1 = p[ro[3] + co[1]]; a2 = p[ro[3] + co[2]]; a3 = p[ro[3] + co[3]];
        BL = fmaf(a0, f[12], BL);  BL = fmaf(a1, f[13], BL);
        BR = fmaf(a2, f[14], BR);  BR = fmaf(a3, f[15], BR);
        dst[(int64_t)c * cs] = blend4(alpha, beta, TL, TR, BL, BR);
    }
}

// one channel of one valid pixel, fs == 4, from precomputed clamped row / column offsets (the body of
// fi4_channels_direct's loop)
__device__ __forceinline__ float fi4_value(const float* __restrict__ p, const unsigned (&ro)[4], const unsigned (&co)[4],
                                           const float (&f)[16], float alpha, float beta) {
    float a0 = p[ro[0] + co[0]], a1 = p[ro[0] + co[1]], a2 = p[ro[0] + co[2]], a3 = p[ro[0] + co[3]];
    float TL = a0 * f[0];  TL = fmaf(a1, f[1], TL);
    float TR = a2 * f[2];  TR = fmaf(a3, f[3], TR);
    a0 = p[ro[1] + co[0]]; a1 = p[ro[1] + co[1]]; a2 = p[ro[1] + co[2]]; a3 = p[ro[1] + co[3]];
    TL = fmaf(a0, f[4], TL);  TL = fmaf(a1, f[5], TL);
    TR = fmaf(a2, f[6], TR);  TR = fmaf(a3, f[7], TR);
    a0 = p[ro[2] + co[0]]; a1 = p[ro[2] + co[1]]; a2 = p[ro[2] + co[2]]; a3 = p[ro[2] + co[3]];
    float BL = a0 * f[8];   BL = fmaf(a1, f[9], BL);
    float BR = a2 * f[10];  BR = fmaf(a3, f[11], BR);
    a0 = p[ro[3] + co[0]]; a1 = p[ro[3] + co[1]]; a2 = p[ro[3] + co[2]]; a3 = p[ro[3] + co[3]];
    BL = fmaf(a0, f[12], BL);  BL = fmaf(a1, f[13], BL);
    BR = fmaf(a2, f[14], BR);  BR = fmaf(a3, f[15], BR);
    return blend4(alpha, beta, TL, TR, BL, BR);
}

// quadrant sums for a runtime filter size, rows outer / columns inner per quadrant
__device__ __forceinline__ void quadrants_generic(const float* __restrict__ plane, const float* __restrict__ fpx,
                                                  int64_t fcs, int hs, int h, int w, int fs,
                                                  int L, int T, int ix, int iy, float q[4]) {
    const int R = L + fs, Bm = T + fs;
    float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
    for (int j = T; j <= iy; ++j) {
        const float* row = plane + (int64_t)clampi(j, 0, h - 1) * hs;
        for (int i = L; i <= ix; ++i)
            TL = fmaf(row[clampi(i, 0, w - 1)], fpx[(int64_t)((j - T) * fs + (i - L)) * fcs], TL);
    }
    for (int j = T; j <= iy; ++j) {
        const float* row = plane + (int64_t)clampi(j, 0, h - 1) * hs;
        for (int i = ix + 1; i < R; ++i)
            TR = fmaf(row[clampi(i, 0, w - 1)], fpx[(int64_t)((j - T) * fs + (i - L)) * fcs], TR);
    }
    for (int j = iy + 1; j < Bm; ++j) {
        const float* row = plane + (int64_t)clampi(j, 0, h - 1) * hs;
        for (int i = L; i <= ix; ++i)
            BL = fmaf(row[clampi(i, 0, w - 1)], fpx[(int64_t)((j - T) * fs + (i - L)) * fcs], BL);
    }
    for (int j = iy + 1; j < Bm; ++j) {
        const float* row = plane + (int64_t)clampi(j, 0, h - 1) * hs;
        for (int i = ix + 1; i < R; ++i)
            BR = fmaf(row[clampi(i, 0, w - 1)], fpx[(int64_t)((j - T) * fs + (i - L)) * fcs], BR);
    }
    q[0] = TL; q[1] = TR; q[2] = BL; q[3] = BR;
}

}  // namespace vfi
