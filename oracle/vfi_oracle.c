/*
 * vfi_oracle.c -- CPU restatement (plain C, fp32) of the reference's hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- see vfi_oracle.h.  "parity unpinned": the
 * reference holds no golden vectors for this path; this file follows the .cu
 * sources operation by operation and is pinned by analytic cases and by an
 * independent numpy formulation (tests/test_oracle.py).
 *
 * Written from the behavioural description of the reference kernels (one
 * output / source pixel per loop iteration, channel loop inside, same operation
 * order).  Compile with -ffp-contract=off: fused multiply-adds appear only
 * where `fmad` asks for them, through fmaf().
 */
#include "vfi_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif


typedef long long i64;

/* threads used by the OpenMP loops of the deterministic per-pixel ops (cpu_baseline timing);
 * scatter ops stay sequential so their summation order is fixed */
static int g_threads = 1;
void vfi_oracle_set_num_threads(int n) { g_threads = n < 1 ? 1 : n; }
int vfi_oracle_get_num_threads(void) { return g_threads; }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int mini(int a, int b) { return a < b ? a : b; }
static inline int maxi(int a, int b) { return a > b ? a : b; }

/* acc + a*b, fused or not */
static inline float mac(float a, float b, float acc, int fmad) {
    if (fmad) return fmaf(a, b, acc);
    float p = a * b;
    return acc + p;
}

/* the four-term bilinear blend every warp op ends with:
 * (1-a)*(1-b)*TL + a*(1-b)*TR + (1-a)*b*BL + a*b*BR, evaluated left to right
 * (filterinterpolation_cuda_kernel.cu:2789-2793, interpolation_cuda_kernel.cu:86-87) */
static inline float blend4(float a, float b, float TL, float TR, float BL, float BR, int fmad) {
    float w00 = (1.0f - a) * (1.0f - b);
    float w10 = a * (1.0f - b);
    float w01 = (1.0f - a) * b;
    float w11 = a * b;
    float t = w00 * TL;
    t = mac(w10, TR, t, fmad);
    t = mac(w01, BL, t, fmad);
    t = mac(w11, BR, t, fmad);
    return t;
}

/* validity test shared by A1, A1b-d and A2 (filterinterpolation_cuda_kernel.cu:2735-2736) */
static inline int fi_valid(float fx, float fy, float x2, float y2, int W, int H) {
    return x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1) &&
           fabsf(fx) < (float)W / 2.0f && fabsf(fy) < (float)H / 2.0f;
}

/* ------------------------------------------------------------------ A1 */

/* the four quadrant sums of one pixel/channel, rows outer, columns inner
 * (filterinterpolation_cuda_kernel.cu:2749-2787) */
static inline void fi_quadrants(const float* plane, const float* filt_px, i64 filt_cstride,
                                int H, int W, int fs, int L, int T, int ix, int iy, int fmad,
                                float q[4]) {
    const int R = L + fs, Bm = T + fs;
    float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
    for (int j = T; j <= iy; ++j) {
        const int cj = clampi(j, 0, H - 1);
        for (int i = L; i <= ix; ++i) {
            const int ci = clampi(i, 0, W - 1);
            TL = mac(plane[(i64)cj * W + ci], filt_px[(i64)((j - T) * fs + (i - L)) * filt_cstride], TL, fmad);
        }
    }
    for (int j = T; j <= iy; ++j) {
        const int cj = clampi(j, 0, H - 1);
        for (int i = ix + 1; i < R; ++i) {
            const int ci = clampi(i, 0, W - 1);
            TR = mac(plane[(i64)cj * W + ci], filt_px[(i64)((j - T) * fs + (i - L)) * filt_cstride], TR, fmad);
        }
    }
    for (int j = iy + 1; j < Bm; ++j) {
        const int cj = clampi(j, 0, H - 1);
        for (int i = L; i <= ix; ++i) {
            const int ci = clampi(i, 0, W - 1);
            BL = mac(plane[(i64)cj * W + ci], filt_px[(i64)((j - T) * fs + (i - L)) * filt_cstride], BL, fmad);
        }
    }
    for (int j = iy + 1; j < Bm; ++j) {
        const int cj = clampi(j, 0, H - 1);
        for (int i = ix + 1; i < R; ++i) {
            const int ci = clampi(i, 0, W - 1);
            BR = mac(plane[(i64)cj * W + ci], filt_px[(i64)((j - T) * fs + (i - L)) * filt_cstride], BR, fmad);
        }
    }
    q[0] = TL; q[1] = TR; q[2] = BL; q[3] = BR;
}

int vfi_oracle_filterinterp_ori_fwd(const float* img, const float* flow, const float* filt,
                                    float* out, int B, int C, int H, int W, int filt_ch,
                                    int fmad, int nthreads) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || filt_ch <= 0) return 1;
    const int fs = (int)sqrtf((float)filt_ch);            /* filterinterpolation_cuda.cc:556-557 */
    const i64 HW = (i64)H * W;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = g_threads;
#pragma omp parallel for collapse(2) schedule(dynamic, 4) num_threads(nthreads)
#else
    (void)nthreads;
#endif
    for (int b = 0; b < B; ++b) {
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const i64 px = (i64)y * W + x;
                const float fx = flow[((i64)b * 2 + 0) * HW + px];
                const float fy = flow[((i64)b * 2 + 1) * HW + px];
                const float x2 = (float)x + fx;
                const float y2 = (float)y + fy;
                if (fi_valid(fx, fy, x2, y2, W, H)) {
                    const int ix = (int)x2, iy = (int)y2;
                    const int L = ix + 1 - fs / 2;
                    const int T = iy + 1 - fs / 2;
                    const float alpha = x2 - (float)ix;
                    const float beta = y2 - (float)iy;
                    const float* fpx = filt + (i64)b * filt_ch * HW + px;
                    for (int c = 0; c < C; ++c) {
                        float q[4];
                        fi_quadrants(img + ((i64)b * C + c) * HW, fpx, HW, H, W, fs, L, T, ix, iy, fmad, q);
                        out[((i64)b * C + c) * HW + px] = blend4(alpha, beta, q[0], q[1], q[2], q[3], fmad);
                    }
                } else {
                    /* copy-through, not zero fill (:2814-2818) */
                    for (int c = 0; c < C; ++c)
                        out[((i64)b * C + c) * HW + px] = img[((i64)b * C + c) * HW + px];
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ A2 */

int vfi_oracle_filterinterp_ori_bwd(const float* img, const float* flow, const float* filt,
                                    const float* gout, float* gimg, float* gflow, float* gfilt,
                                    int B, int C, int H, int W, int filt_ch, int fmad) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || filt_ch <= 0) return 1;
    const int fs = (int)sqrtf((float)filt_ch);
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (!fi_valid(fx, fy, x2, y2, W, H)) continue;      /* no gradient at all (:2863-2864) */
        const int ix = (int)x2, iy = (int)y2;
        const int L = ix + 1 - fs / 2, T = iy + 1 - fs / 2;
        const int R = L + fs, Bm = T + fs;
        const float alpha = x2 - (float)ix;
        const float beta = y2 - (float)iy;
        const float* fpx = filt + (i64)b * filt_ch * HW + px;
        float* gfpx = gfilt + (i64)b * filt_ch * HW + px;
        /* steps 1+3: image and filter gradients, scattered (:2881-2944) */
        for (int c = 0; c < C; ++c) {
            const float* plane = img + ((i64)b * C + c) * HW;
            float* gplane = gimg + ((i64)b * C + c) * HW;
            const float g = gout[((i64)b * C + c) * HW + px];
            const float qg[4] = { g * (1.0f - alpha) * (1.0f - beta), g * alpha * (1.0f - beta),
                                  g * (1.0f - alpha) * beta,          g * alpha * beta };
            for (int quad = 0; quad < 4; ++quad) {
                const int j0 = (quad < 2) ? T : iy + 1, j1 = (quad < 2) ? iy : Bm - 1;
                const int i0 = (quad & 1) ? ix + 1 : L, i1 = (quad & 1) ? R - 1 : ix;
                for (int j = j0; j <= j1; ++j) {
                    const int cj = clampi(j, 0, H - 1);
                    for (int i = i0; i <= i1; ++i) {
                        const int ci = clampi(i, 0, W - 1);
                        const i64 k = (i64)((j - T) * fs + (i - L)) * HW;
                        gplane[(i64)cj * W + ci] += qg[quad] * fpx[k];
                        gfpx[k] += qg[quad] * plane[(i64)cj * W + ci];
                    }
                }
            }
        }
        /* step 2: flow gradient by quadrant differences (:2965-3102) */
        float gx = 0.0f, gy = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float g = gout[((i64)b * C + c) * HW + px];
            float q[4];
            fi_quadrants(img + ((i64)b * C + c) * HW, fpx, HW, H, W, fs, L, T, ix, iy, fmad, q);
            {
                const float gamma = 1.0f - beta;
                float temp = 0.0f;
                temp = mac(gamma, q[1] - q[0], temp, fmad);
                temp = mac(1.0f - gamma, q[3] - q[2], temp, fmad);
                gx = mac(g, temp, gx, fmad);
            }
            {
                const float gamma = 1.0f - alpha;
                float temp = 0.0f;
                temp = mac(gamma, q[2] - q[0], temp, fmad);
                temp = mac(1.0f - gamma, q[3] - q[1], temp, fmad);
                gy = mac(g, temp, gy, fmad);
            }
        }
        gflow[((i64)b * 2 + 0) * HW + px] = gx;
        gflow[((i64)b * 2 + 1) * HW + px] = gy;
    }
    return 0;
}

/* ------------------------------------------------------- A1b / A1c / A1d */

/* one displaced tap: bilinear sample of `plane` at (clamped tap + learned offset)
 * (filterinterpolation_cuda_kernel.cu:98-111).  The reference does not clamp the
 * four corner indices (out-of-bounds reads when the displaced tap leaves the
 * image: undefined there); this restatement clamps them to the image, which
 * leaves every in-range result unchanged. */
static inline float defor_tap(const float* plane, int H, int W, float fracY, float fracX, int fmad) {
    const int Top = (int)fracY, Left = (int)fracX;
    const float phiY = fracY - (float)Top;
    const float phiX = fracX - (float)Left;
    const int Bottom = Top + 1, Right = Left + 1;
    const float PTL = (1.0f - phiX) * (1.0f - phiY);
    const float PTR = phiX * (1.0f - phiY);
    const float PBL = (1.0f - phiX) * phiY;
    const float PBR = phiY * phiX;
    const int t = clampi(Top, 0, H - 1), bo = clampi(Bottom, 0, H - 1);
    const int l = clampi(Left, 0, W - 1), r = clampi(Right, 0, W - 1);
    float s = PTL * plane[(i64)t * W + l];
    s = mac(PTR, plane[(i64)t * W + r], s, fmad);
    s = mac(PBL, plane[(i64)bo * W + l], s, fmad);
    s = mac(PBR, plane[(i64)bo * W + r], s, fmad);
    return s;
}

int vfi_oracle_filterinterp_defor_fwd(int variant, const float* img, const float* flow,
                                      const float* filt, const float* off, float* out,
                                      int B, int C, int H, int W, int fs, int fmad) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || fs <= 0 || variant < 0 || variant > 2) return 1;
    const i64 HW = (i64)H * W;
    const int fs2 = fs * fs;
    /* 4-input forward only has a body for fs 4 and 6; other sizes leave the
     * caller's zero-filled output untouched (:68) */
    if (variant == 0 && !(fs == 4 || fs == 6)) return 0;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (!fi_valid(fx, fy, x2, y2, W, H)) {
            for (int c = 0; c < C; ++c)
                out[((i64)b * C + c) * HW + px] = img[((i64)b * C + c) * HW + px];
            continue;
        }
        const int ix = (int)x2, iy = (int)y2;
        const int L = ix + 1 - fs / 2, T = iy + 1 - fs / 2;
        const int R = L + fs, Bm = T + fs;
        const float alpha = x2 - (float)ix;
        const float beta = y2 - (float)iy;
        const float* opx = off + (i64)b * 2 * fs2 * HW + px;
        const float* fpx = (variant == 2) ? NULL : filt + (i64)b * fs2 * HW + px;
        for (int c = 0; c < C; ++c) {
            const float* plane = img + ((i64)b * C + c) * HW;
            float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
            if (variant == 0) {
                /* quadrant membership by integer loops, TL, TR, BL, BR in turn (:90-198) */
                for (int quad = 0; quad < 4; ++quad) {
                    const int j0 = (quad < 2) ? T : iy + 1, j1 = (quad < 2) ? iy : Bm - 1;
                    const int i0 = (quad & 1) ? ix + 1 : L, i1 = (quad & 1) ? R - 1 : ix;
                    float acc = 0.0f;
                    for (int j = j0; j <= j1; ++j) {
                        const int cj = clampi(j, 0, H - 1);
                        for (int i = i0; i <= i1; ++i) {
                            const int ci = clampi(i, 0, W - 1);
                            const int k = (j - T) * fs + (i - L);
                            const float fracY = (float)cj + opx[(i64)k * HW];
                            const float fracX = (float)ci + opx[(i64)(fs2 + k) * HW];
                            acc = mac(defor_tap(plane, H, W, fracY, fracX, fmad), fpx[(i64)k * HW], acc, fmad);
                        }
                    }
                    if (quad == 0) TL = acc; else if (quad == 1) TR = acc; else if (quad == 2) BL = acc; else BR = acc;
                }
            } else {
                /* one sweep over the window; quadrant by displaced position (:1401-1471, :2118-2162) */
                for (int j = T; j < Bm; ++j) {
                    const int cj = clampi(j, 0, H - 1);
                    for (int i = L; i < R; ++i) {
                        const int ci = clampi(i, 0, W - 1);
                        const int k = (j - T) * fs + (i - L);
                        const float fracY = (float)cj + opx[(i64)k * HW];
                        const float fracX = (float)ci + opx[(i64)(fs2 + k) * HW];
                        const float v = defor_tap(plane, H, W, fracY, fracX, fmad);
                        const float wgt = (variant == 2) ? 1.0f : fpx[(i64)k * HW];
                        if (fracX <= x2 && fracY <= y2) TL = (variant == 2) ? TL + v : mac(v, wgt, TL, fmad);
                        if (fracX >  x2 && fracY <= y2) TR = (variant == 2) ? TR + v : mac(v, wgt, TR, fmad);
                        if (fracX <= x2 && fracY >  y2) BL = (variant == 2) ? BL + v : mac(v, wgt, BL, fmad);
                        if (fracX >  x2 && fracY >  y2) BR = (variant == 2) ? BR + v : mac(v, wgt, BR, fmad);
                    }
                }
            }
            out[((i64)b * C + c) * HW + px] = blend4(alpha, beta, TL, TR, BL, BR, fmad);
        }
    }
    return 0;
}

/* backward of the three deformable variants (filterinterpolation_cuda_kernel.cu:430-1215,
 * 1500-1935, 2195-2567).  Per valid pixel and tap k: quadrant weight Wq chosen by integer index
 * (variant 0) or by displaced position (variants 1, 2);
 *   gimg[clamped UNDISPLACED tap] += g*Wq * filt[k]      (the reference's own approximation)
 *   gfilt[k]                      += g*Wq * tap_value
 *   goff_y[k], goff_x[k]          += g*Wq * d(tap_value)/d(offset) * filt[k]
 *   gflow = quadrant differences of the forward sums, as in the _ori backward.
 * Variant 2 has no filter: filt == NULL, gfilt == NULL, weights are 1. */
int vfi_oracle_filterinterp_defor_bwd(int variant, const float* img, const float* flow,
                                      const float* filt, const float* off, const float* gout,
                                      float* gimg, float* gflow, float* gfilt, float* goff,
                                      int B, int C, int H, int W, int fs, int fmad) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || fs <= 0 || variant < 0 || variant > 2) return 1;
    const i64 HW = (i64)H * W;
    const int fs2 = fs * fs;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (!fi_valid(fx, fy, x2, y2, W, H)) continue;
        const int ix = (int)x2, iy = (int)y2;
        const int L = ix + 1 - fs / 2, T = iy + 1 - fs / 2;
        const float alpha = x2 - (float)ix;
        const float beta = y2 - (float)iy;
        const float* opx = off + (i64)b * 2 * fs2 * HW + px;
        const float* fpx = (variant == 2) ? NULL : filt + (i64)b * fs2 * HW + px;
        float* gfpx = (variant == 2) ? NULL : gfilt + (i64)b * fs2 * HW + px;
        float* gopx = goff + (i64)b * 2 * fs2 * HW + px;
        const float kq[4] = { (1.0f - alpha) * (1.0f - beta), alpha * (1.0f - beta),
                              (1.0f - alpha) * beta,          alpha * beta };
        float gx = 0.0f, gy = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float* plane = img + ((i64)b * C + c) * HW;
            float* gplane = gimg + ((i64)b * C + c) * HW;
            const float g = gout[((i64)b * C + c) * HW + px];
            const float qg[4] = { g * (1.0f - alpha) * (1.0f - beta), g * alpha * (1.0f - beta),
                                  g * (1.0f - alpha) * beta,          g * alpha * beta };
            float q[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int dj = 0; dj < fs; ++dj) {
                const int j = T + dj, cj = clampi(j, 0, H - 1);
                for (int di = 0; di < fs; ++di) {
                    const int i = L + di, ci = clampi(i, 0, W - 1);
                    const int k = dj * fs + di;
                    const float fracY = (float)cj + opx[(i64)k * HW];
                    const float fracX = (float)ci + opx[(i64)(fs2 + k) * HW];
                    int quad;
                    if (variant == 0) quad = (j > iy ? 2 : 0) + (i > ix ? 1 : 0);
                    else if (fracX <= x2 && fracY <= y2) quad = 0;
                    else if (fracX > x2 && fracY <= y2) quad = 1;
                    else if (fracX <= x2 && fracY > y2) quad = 2;
                    else if (fracX > x2 && fracY > y2) quad = 3;
                    else continue;                              /* NaN position: no quadrant */
                    const float wgt = (variant == 2) ? 1.0f : fpx[(i64)k * HW];
                    /* tap value and its derivatives w.r.t. the offsets; corners clamped as in the forward */
                    const int Top = (int)fracY, Left = (int)fracX;
                    const float phiY = fracY - (float)Top, phiX = fracX - (float)Left;
                    const int t = clampi(Top, 0, H - 1), bo = clampi(Top + 1, 0, H - 1);
                    const int l = clampi(Left, 0, W - 1), r = clampi(Left + 1, 0, W - 1);
                    const float vTL = plane[(i64)t * W + l], vTR = plane[(i64)t * W + r];
                    const float vBL = plane[(i64)bo * W + l], vBR = plane[(i64)bo * W + r];
                    float v = ((1.0f - phiX) * (1.0f - phiY)) * vTL;
                    v = mac(phiX * (1.0f - phiY), vTR, v, fmad);
                    v = mac((1.0f - phiX) * phiY, vBL, v, fmad);
                    v = mac(phiY * phiX, vBR, v, fmad);
                    float dY = (-(1.0f - phiX)) * vTL;          /* - (1-phiX) TL + (1-phiX) BL - phiX TR + phiX BR */
                    dY = mac(1.0f - phiX, vBL, dY, fmad);
                    dY = mac(-phiX, vTR, dY, fmad);
                    dY = mac(phiX, vBR, dY, fmad);
                    float dX = (-(1.0f - phiY)) * vTL;          /* - (1-phiY) TL + (1-phiY) TR - phiY BL + phiY BR */
                    dX = mac(1.0f - phiY, vTR, dX, fmad);
                    dX = mac(-phiY, vBL, dX, fmad);
                    dX = mac(phiY, vBR, dX, fmad);
                    if (variant == 2) {
                        gplane[(i64)cj * W + ci] += qg[quad];
                        q[quad] = q[quad] + v;
                        gopx[(i64)k * HW] += g * kq[quad] * dY;
                        gopx[(i64)(fs2 + k) * HW] += g * kq[quad] * dX;
                    } else {
                        gplane[(i64)cj * W + ci] += qg[quad] * wgt;
                        gfpx[(i64)k * HW] += qg[quad] * v;
                        q[quad] = mac(v, wgt, q[quad], fmad);
                        gopx[(i64)k * HW] += g * kq[quad] * dY * wgt;
                        gopx[(i64)(fs2 + k) * HW] += g * kq[quad] * dX * wgt;
                    }
                }
            }
            {
                const float gamma = 1.0f - beta;
                float temp = 0.0f;
                temp = mac(gamma, q[1] - q[0], temp, fmad);
                temp = mac(1.0f - gamma, q[3] - q[2], temp, fmad);
                gx = mac(g, temp, gx, fmad);
            }
            {
                const float gamma = 1.0f - alpha;
                float temp = 0.0f;
                temp = mac(gamma, q[2] - q[0], temp, fmad);
                temp = mac(1.0f - gamma, q[3] - q[1], temp, fmad);
                gy = mac(g, temp, gy, fmad);
            }
        }
        gflow[((i64)b * 2 + 0) * HW + px] = gx;
        gflow[((i64)b * 2 + 1) * HW + px] = gy;
    }
    return 0;
}

/* ------------------------------------------------------------ A3 / A4 */

/* pass 3 of both projections (flowprojection_cuda_kernel.cu:175-231,
 * depthflowprojection_cuda_kernel.cu:181-237).  In place: a cell read here is
 * either a non-hole (never written in this pass) or is multiplied by 0. */
static void fillhole_pass(float* count, float* out, int B, int H, int W) {
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b) {
        const float* cnt = count + (i64)b * HW;
        float* o0 = out + ((i64)b * 2 + 0) * HW;
        float* o1 = out + ((i64)b * 2 + 1) * HW;
        for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (!(cnt[(i64)y * W + x] <= 0.0f)) continue;
            int lo = x; float lt = 0.0f;
            while (lt == 0.0f && lo - 1 >= 0) { lo -= 1; lt = cnt[(i64)y * W + lo]; }
            int ro = x; float rt = 0.0f;
            while (rt == 0.0f && ro + 1 <= W - 1) { ro += 1; rt = cnt[(i64)y * W + ro]; }
            int uo = y; float ut = 0.0f;
            while (ut == 0.0f && uo - 1 >= 0) { uo -= 1; ut = cnt[(i64)uo * W + x]; }
            int dn = y; float dt = 0.0f;
            while (dt == 0.0f && dn + 1 <= H - 1) { dn += 1; dt = cnt[(i64)dn * W + x]; }
            if (lt + rt + ut + dt <= 0.0f) continue;
            lt = (lt > 0.0f) ? 1.0f : 0.0f;
            rt = (rt > 0.0f) ? 1.0f : 0.0f;
            ut = (ut > 0.0f) ? 1.0f : 0.0f;
            dt = (dt > 0.0f) ? 1.0f : 0.0f;
            const float den = lt + rt + ut + dt;
            /* products and sums left to right; 0/1 weights make contraction irrelevant */
            o0[(i64)y * W + x] = (lt * o0[(i64)y * W + lo] + rt * o0[(i64)y * W + ro] +
                                  ut * o0[(i64)uo * W + x] + dt * o0[(i64)dn * W + x]) / den;
            o1[(i64)y * W + x] = (lt * o1[(i64)y * W + lo] + rt * o1[(i64)y * W + ro] +
                                  ut * o1[(i64)uo * W + x] + dt * o1[(i64)dn * W + x]) / den;
        }
    }
}

static void average_pass(const float* count, float* out, int B, int H, int W) {
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
        for (i64 p = 0; p < HW; ++p) {
            const float c = count[(i64)b * HW + p];
            if (c > 0.0f) {
                out[((i64)b * 2 + 0) * HW + p] /= c;
                out[((i64)b * 2 + 1) * HW + p] /= c;
            }
        }
}

static int project(const float* flow, const float* depth, float* count, float* out,
                   int B, int H, int W, int fillhole) {
    if (B <= 0 || H <= 0 || W <= 0) return 1;
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1))) continue;
        const int L = (int)x2, T = (int)y2;
        const int R = mini(L + 1, W - 1), Bm = mini(T + 1, H - 1);
        float ax, ay, ac;
        if (depth) {                        /* depthflowprojection_cuda_kernel.cu:74-91 */
            const float d = depth[(i64)b * HW + px];
            ax = -d * fx; ay = -d * fy; ac = d * 1.0f;
        } else {                            /* flowprojection_cuda_kernel.cu:75-88 */
            ax = -fx; ay = -fy; ac = 1.0f;
        }
        const i64 t[4] = { (i64)T * W + L, (i64)T * W + R, (i64)Bm * W + L, (i64)Bm * W + R };
        for (int k = 0; k < 4; ++k) {       /* R==L / Bm==T at the far edges: same cell twice */
            out[((i64)b * 2 + 0) * HW + t[k]] += ax;
            out[((i64)b * 2 + 1) * HW + t[k]] += ay;
            count[(i64)b * HW + t[k]] += ac;
        }
    }
    average_pass(count, out, B, H, W);
    if (fillhole) fillhole_pass(count, out, B, H, W);
    return 0;
}

int vfi_oracle_flowproj_fwd(const float* flow, float* count, float* out,
                            int B, int H, int W, int fillhole) {
    return project(flow, NULL, count, out, B, H, W, fillhole);
}

int vfi_oracle_depthflowproj_fwd(const float* flow, const float* depth, float* count, float* out,
                                 int B, int H, int W, int fillhole, int fmad) {
    (void)fmad;   /* every product feeds an atomic add: nothing to contract */
    return project(flow, depth, count, out, B, H, W, fillhole);
}

/* MinDepthFlowProjection: mindepthflowprojection_cuda_kernel.cu:27-119 executed sequentially in
 * raster order of the source pixels (the reference's parallel read-compare-write is a race; the
 * sequential order is the library's defined result), then the hole filling of :121-206. */
int vfi_oracle_mindepthflowproj_fwd(const float* flow, const float* weight, float* count, float* out,
                                    int B, int H, int W, int fillhole) {
    if (B <= 0 || H <= 0 || W <= 0) return 1;
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b) {
        float* o0 = out + ((i64)b * 2 + 0) * HW;
        float* o1 = out + ((i64)b * 2 + 1) * HW;
        float* cn = count + (i64)b * HW;
        for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const i64 px = (i64)y * W + x;
            const float fx = flow[((i64)b * 2 + 0) * HW + px];
            const float fy = flow[((i64)b * 2 + 1) * HW + px];
            const float x2 = (float)x + fx;
            const float y2 = (float)y + fy;
            if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1))) continue;   /* (:68) */
            const i64 t = (i64)(int)y2 * W + (int)x2;        /* top-left target only (:69-70, 76-84) */
            const float wgt = weight[(i64)b * HW + px];
            if (wgt > cn[t]) {
                o0[t] = -fx;
                o1[t] = -fy;
                cn[t] = wgt;
            }
        }
        if (!fillhole) continue;
        /* holes read non-holes only, and only holes are written: any visiting order gives the same result */
        for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            if (!(cn[(i64)y * W + x] <= 0.0f)) continue;
            int lo = x; float lt = 0.0f;
            while (lt == 0.0f && lo - 1 >= 0) { --lo; lt = cn[(i64)y * W + lo]; }
            int ro = x; float rt = 0.0f;
            while (rt == 0.0f && ro + 1 <= W - 1) { ++ro; rt = cn[(i64)y * W + ro]; }
            int uo = y; float ut = 0.0f;
            while (ut == 0.0f && uo - 1 >= 0) { --uo; ut = cn[(i64)uo * W + x]; }
            int dn = y; float dt = 0.0f;
            while (dt == 0.0f && dn + 1 <= H - 1) { ++dn; dt = cn[(i64)dn * W + x]; }
            if (lt + rt + ut + dt <= 0.0f) continue;
            lt = lt > 0.0f ? 1.0f : 0.0f;  rt = rt > 0.0f ? 1.0f : 0.0f;
            ut = ut > 0.0f ? 1.0f : 0.0f;  dt = dt > 0.0f ? 1.0f : 0.0f;
            float* po[2] = { o0, o1 };
            for (int ch = 0; ch < 2; ++ch) {                 /* weights are 0/1: products exact, fmad irrelevant */
                const float* p = po[ch];
                const float acc = lt * p[(i64)y * W + lo] + rt * p[(i64)y * W + ro] + ut * p[(i64)uo * W + x] +
                                  dt * p[(i64)dn * W + x];
                po[ch][(i64)y * W + x] = acc / (lt + rt + ut + dt);
            }
        }
    }
    return 0;
}

/* mindepthflowprojection_cuda_kernel.cu:209-331: flow gradient only (the weight gradient is commented out there) */
int vfi_oracle_mindepthflowproj_bwd(const float* flow, const float* weight, const float* count, const float* gout,
                                    float* gflow, int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 1;
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1))) continue;
        const int L = (int)x2, T = (int)y2;
        const int R = mini(L + 1, W - 1), Bm = mini(T + 1, H - 1);
        const float wgt = weight[(i64)b * HW + px];
        const i64 t[4] = { (i64)T * W + L, (i64)T * W + R, (i64)Bm * W + L, (i64)Bm * W + R };
        for (int k = 0; k < 4; ++k)                          /* (:271-286) */
            if (wgt == count[(i64)b * HW + t[k]]) {
                gflow[((i64)b * 2 + 0) * HW + px] += -gout[((i64)b * 2 + 0) * HW + t[k]];
                gflow[((i64)b * 2 + 1) * HW + px] += -gout[((i64)b * 2 + 1) * HW + t[k]];
            }
    }
    return 0;
}

int vfi_oracle_flowproj_bwd(const float* flow, const float* count, const float* gout,
                            float* gflow, int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 1;
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1))) continue;
        const int L = (int)x2, T = (int)y2;
        const int R = mini(L + 1, W - 1), Bm = mini(T + 1, H - 1);
        const i64 t[4] = { (i64)T * W + L, (i64)T * W + R, (i64)Bm * W + L, (i64)Bm * W + R };
        for (int ch = 0; ch < 2; ++ch) {
            float g = gflow[((i64)b * 2 + ch) * HW + px];
            for (int k = 0; k < 4; ++k)      /* g += -(gout/count) (:279-296) */
                g += -gout[((i64)b * 2 + ch) * HW + t[k]] / count[(i64)b * HW + t[k]];
            gflow[((i64)b * 2 + ch) * HW + px] = g;
        }
    }
    return 0;
}

int vfi_oracle_depthflowproj_bwd(const float* flow, const float* depth, const float* count,
                                 const float* out, const float* gout, float* gflow, float* gdepth,
                                 int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 1;
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (!(x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1))) continue;
        const int L = (int)x2, T = (int)y2;
        const int R = mini(L + 1, W - 1), Bm = mini(T + 1, H - 1);
        const float d = depth[(i64)b * HW + px];
        const i64 t[4] = { (i64)T * W + L, (i64)T * W + R, (i64)Bm * W + L, (i64)Bm * W + R };
        for (int ch = 0; ch < 2; ++ch) {     /* (:291-311) */
            float g = gflow[((i64)b * 2 + ch) * HW + px];
            for (int k = 0; k < 4; ++k)
                g += -gout[((i64)b * 2 + ch) * HW + t[k]] * d / count[(i64)b * HW + t[k]];
            gflow[((i64)b * 2 + ch) * HW + px] = g;
        }
        float gd = gdepth[(i64)b * HW + px]; /* (:314-336) */
        for (int ch = 0; ch < 2; ++ch) {
            const float f = ch ? fy : fx;
            for (int k = 0; k < 4; ++k)
                gd += -gout[((i64)b * 2 + ch) * HW + t[k]] / count[(i64)b * HW + t[k]] *
                      (f - out[((i64)b * 2 + ch) * HW + t[k]]);
        }
        gdepth[(i64)b * HW + px] = gd;
    }
    return 0;
}

/* ------------------------------------------------------------------ A5 */

int vfi_oracle_interp_fwd(const float* img, const float* flow, float* out,
                          int B, int C, int H, int W, int fmad) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (x2 >= 0.0f && y2 >= 0.0f && x2 < (float)W && y2 < (float)H) {   /* strict upper bound (:71) */
            const int L = (int)x2, T = (int)y2;
            const int R = mini(L + 1, W - 1), Bm = mini(T + 1, H - 1);
            const float alpha = x2 - (float)L, beta = y2 - (float)T;
            for (int c = 0; c < C; ++c) {
                const float* p = img + ((i64)b * C + c) * HW;
                out[((i64)b * C + c) * HW + px] =
                    blend4(alpha, beta, p[(i64)T * W + L], p[(i64)T * W + R],
                           p[(i64)Bm * W + L], p[(i64)Bm * W + R], fmad);
            }
        } else {
            for (int c = 0; c < C; ++c) out[((i64)b * C + c) * HW + px] = 0.0f;
        }
    }
    return 0;
}

int vfi_oracle_interp_bwd(const float* img, const float* flow, const float* gout,
                          float* gimg, float* gflow, int B, int C, int H, int W, int fmad) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const i64 px = (i64)y * W + x;
        const float fx = flow[((i64)b * 2 + 0) * HW + px];
        const float fy = flow[((i64)b * 2 + 1) * HW + px];
        const float x2 = (float)x + fx;
        const float y2 = (float)y + fy;
        if (!(x2 >= 0.0f && y2 >= 0.0f && x2 < (float)W && y2 < (float)H)) continue;
        const int L = (int)x2, T = (int)y2;
        const int R = mini(L + 1, W - 1), Bm = mini(T + 1, H - 1);
        const float alpha = x2 - (float)L, beta = y2 - (float)T;
        for (int c = 0; c < C; ++c) {          /* (:151-158) */
            float* gp = gimg + ((i64)b * C + c) * HW;
            const float g = gout[((i64)b * C + c) * HW + px];
            gp[(i64)T * W + L]  += g * (1.0f - alpha) * (1.0f - beta);
            gp[(i64)T * W + R]  += g * alpha * (1.0f - beta);
            gp[(i64)Bm * W + L] += g * (1.0f - alpha) * beta;
            gp[(i64)Bm * W + R] += g * alpha * beta;
        }
        float gamma = (float)Bm - y2;           /* (:161-176) */
        float bot = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float* p = img + ((i64)b * C + c) * HW;
            float temp = 0.0f;
            temp = mac(gamma, p[(i64)T * W + R] - p[(i64)T * W + L], temp, fmad);
            temp = mac(1.0f - gamma, p[(i64)Bm * W + R] - p[(i64)Bm * W + L], temp, fmad);
            bot = mac(gout[((i64)b * C + c) * HW + px], temp, bot, fmad);
        }
        gflow[((i64)b * 2 + 0) * HW + px] = bot;
        gamma = (float)R - x2;                  /* (:181-196) */
        bot = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float* p = img + ((i64)b * C + c) * HW;
            float temp = 0.0f;
            temp = mac(gamma, p[(i64)Bm * W + L] - p[(i64)T * W + L], temp, fmad);
            temp = mac(1.0f - gamma, p[(i64)Bm * W + R] - p[(i64)T * W + R], temp, fmad);
            bot = mac(gout[((i64)b * C + c) * HW + px], temp, bot, fmad);
        }
        gflow[((i64)b * 2 + 1) * HW + px] = bot;
    }
    return 0;
}

/* ------------------------------------------------------------------ A6 */

int vfi_oracle_sepconv_fwd(const float* img, const float* v, const float* h, float* out,
                           int B, int C, int H, int W, int fs, int fmad) {
    const int oH = H - fs + 1, oW = W - fs + 1;
    if (B <= 0 || C <= 0 || fs <= 0 || oH <= 0 || oW <= 0) return 1;
    const i64 HW = (i64)H * W, oHW = (i64)oH * oW;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < oH; ++y)
    for (int x = 0; x < oW; ++x) {
        const i64 px = (i64)y * oW + x;
        for (int c = 0; c < C; ++c) {
            const float* p = img + ((i64)b * C + c) * HW;
            float acc = 0.0f;
            for (int fy = 0; fy < fs; ++fy)
            for (int fx = 0; fx < fs; ++fx) {
                const float t1 = p[(i64)(y + fy) * W + (x + fx)];
                const float t2 = v[((i64)b * fs + fy) * oHW + px];
                const float t3 = h[((i64)b * fs + fx) * oHW + px];
                acc = mac(t1 * t2, t3, acc, fmad);      /* out += t1*t2*t3 (:72) */
            }
            out[((i64)b * C + c) * oHW + px] = acc;
        }
    }
    return 0;
}

int vfi_oracle_sepconv_bwd(const float* img, const float* v, const float* h, const float* gout,
                           float* gimg, float* gv, float* gh,
                           int B, int C, int H, int W, int fs) {
    const int oH = H - fs + 1, oW = W - fs + 1;
    if (B <= 0 || C <= 0 || fs <= 0 || oH <= 0 || oW <= 0) return 1;
    const i64 HW = (i64)H * W, oHW = (i64)oH * oW;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < oH; ++y)
    for (int x = 0; x < oW; ++x) {
        const i64 px = (i64)y * oW + x;
        for (int c = 0; c < C; ++c) {
            const float* p = img + ((i64)b * C + c) * HW;
            float* gp = gimg + ((i64)b * C + c) * HW;
            const float g = gout[((i64)b * C + c) * oHW + px];
            for (int fy = 0; fy < fs; ++fy)
            for (int fx = 0; fx < fs; ++fx) {       /* (:114-127) */
                const float t1 = p[(i64)(y + fy) * W + (x + fx)];
                const float t2 = v[((i64)b * fs + fy) * oHW + px];
                const float t3 = h[((i64)b * fs + fx) * oHW + px];
                gp[(i64)(y + fy) * W + (x + fx)] += g * t2 * t3;
                gv[((i64)b * fs + fy) * oHW + px] += g * t1 * t3;
                gh[((i64)b * fs + fx) * oHW + px] += g * t1 * t2;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ A7 */

int vfi_oracle_sepconvflow_fwd(const float* v, const float* h, float* flow_out,
                               int B, int H, int W, int fs, int fmad) {
    const int oH = H - fs + 1, oW = W - fs + 1;
    if (B <= 0 || fs <= 0 || oH <= 0 || oW <= 0) return 1;
    const i64 oHW = (i64)oH * oW;
    const double centre = ((double)(float)fs - 1.0) / 2.0;   /* double arithmetic in the source (:75) */
    for (int b = 0; b < B; ++b)
    for (i64 px = 0; px < oHW; ++px) {
        float fy = 0.0f, sy = 0.0f;
        for (int f = 0; f < fs; ++f) {
            const float t = v[((i64)b * fs + f) * oHW + px];
            fy = mac((float)f, t, fy, fmad);
            sy += t;
        }
        fy = (float)((double)(fy / sy) - centre);
        flow_out[((i64)b * 2 + 1) * oHW + px] = (fabsf(sy) > 0.0f) ? fy : -2000.0f;
        float fx = 0.0f, sx = 0.0f;
        for (int f = 0; f < fs; ++f) {
            const float t = h[((i64)b * fs + f) * oHW + px];
            fx = mac((float)f, t, fx, fmad);
            sx += t;
        }
        fx = (float)((double)(fx / sx) - centre);
        flow_out[((i64)b * 2 + 0) * oHW + px] = (fabsf(sx) > 0.0f) ? fx : -2000.0f;
    }
    return 0;
}

int vfi_oracle_sepconvflow_bwd(const float* v, const float* h, const float* gflow,
                               float* gv, float* gh, int B, int H, int W, int fs, int fmad) {
    const int oH = H - fs + 1, oW = W - fs + 1;
    if (B <= 0 || fs <= 0 || oH <= 0 || oW <= 0) return 1;
    const i64 oHW = (i64)oH * oW;
    for (int b = 0; b < B; ++b)
    for (i64 px = 0; px < oHW; ++px) {
        float fy = 0.0f, sy = 0.0f;
        for (int f = 0; f < fs; ++f) {
            const float t = v[((i64)b * fs + f) * oHW + px];
            fy = mac((float)f, t, fy, fmad);
            sy += t;
        }
        if (fabsf(sy) > 0.0f) {                 /* plain store (:144) */
            const float g = gflow[((i64)b * 2 + 1) * oHW + px];
            const float offset = fy / (sy * sy);
            for (int f = 0; f < fs; ++f)
                gv[((i64)b * fs + f) * oHW + px] = g * ((float)f / sy - offset);
        }
        float fx = 0.0f, sx = 0.0f;
        for (int f = 0; f < fs; ++f) {
            const float t = h[((i64)b * fs + f) * oHW + px];
            fx = mac((float)f, t, fx, fmad);
            sx += t;
        }
        if (fabsf(sx) > 0.0f) {                 /* accumulate (:166) */
            const float g = gflow[((i64)b * 2 + 0) * oHW + px];
            const float offset = fx / (sx * sx);
            for (int f = 0; f < fs; ++f) {
                const i64 o = ((i64)b * fs + f) * oHW + px;
                gh[o] = mac(g, (float)f / sx - offset, gh[o], fmad);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ A8 */

int vfi_oracle_correlation_out_dims(int H, int W, int pad, int k, int md, int s1, int s2,
                                    int* outC, int* outH, int* outW) {
    if (k <= 0 || s1 <= 0 || s2 <= 0 || md < 0 || pad < 0) return 1;
    const int kr = (k - 1) / 2;
    const int border = kr + md;
    const int pH = H + 2 * pad, pW = W + 2 * pad;
    const int dr = md / s2;
    *outC = (dr * 2 + 1) * (dr * 2 + 1);
    /* ceil(float / float) exactly as the binding computes it (correlation_cuda.cc:31-32) */
    *outH = (int)ceilf((float)(pH - 2 * border) / (float)s1);
    *outW = (int)ceilf((float)(pW - 2 * border) / (float)s1);
    return 0;
}

/* value of the zero-padded NHWC repack at padded position (yp, xp)
 * (correlation_cuda_kernel.cu:66-69 writes it, cc:37-39 zero-fills it) */
static inline float padded(const float* f, int C, int H, int W, int pad, int b, int c, int yp, int xp) {
    const int y = yp - pad, x = xp - pad;
    if (y < 0 || y >= H || x < 0 || x >= W) return 0.0f;
    return f[(((i64)b * C + c) * H + y) * W + x];
}

int vfi_oracle_correlation_fwd(const float* f1, const float* f2, float* out,
                               int B, int C, int H, int W,
                               int pad, int k, int md, int s1, int s2, int order, int fmad) {
    int oC, oH, oW;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
    if (vfi_oracle_correlation_out_dims(H, W, pad, k, md, s1, s2, &oC, &oH, &oW)) return 1;
    if (oH <= 0 || oW <= 0) return 1;
    const int kr = (k - 1) / 2;
    const int dr = md / s2;
    const int dsz = 2 * dr + 1;
    const int nelems = k * k * C;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(dynamic, 2) num_threads(g_threads)
#endif
    for (int b = 0; b < B; ++b)
    for (int oy = 0; oy < oH; ++oy)
    for (int ox = 0; ox < oW; ++ox) {
        const int y1 = oy * s1 + md, x1 = ox * s1 + md;
        for (int tj = -dr; tj <= dr; ++tj)
        for (int ti = -dr; ti <= dr; ++ti) {
            const int x2 = x1 + ti * s2, y2 = y1 + tj * s2;
            float total;
            if (order == 0) {
                /* 32 lanes, lane l owns channels l, l+32, ...; then the 16-8-4-2-1
                 * shuffle-down tree whose lane 0 is the result (:17-21, :112-131) */
                float lane[32];
                for (int l = 0; l < 32; ++l) {
                    float acc = 0.0f;
                    for (int j = -kr; j <= kr; ++j)
                    for (int i = -kr; i <= kr; ++i)
                    for (int ch = l; ch < C; ch += 32)
                        acc = mac(padded(f1, C, H, W, pad, b, ch, y1 + j, x1 + i),
                                  padded(f2, C, H, W, pad, b, ch, y2 + j, x2 + i), acc, fmad);
                    lane[l] = acc;
                }
                for (int offset = 16; offset > 0; offset /= 2)
                    for (int l = 0; l < offset; ++l) lane[l] += lane[l + offset];
                total = lane[0];
            } else {
                float acc = 0.0f;
                for (int j = -kr; j <= kr; ++j)
                for (int i = -kr; i <= kr; ++i)
                for (int ch = 0; ch < C; ++ch)
                    acc = mac(padded(f1, C, H, W, pad, b, ch, y1 + j, x1 + i),
                              padded(f2, C, H, W, pad, b, ch, y2 + j, x2 + i), acc, fmad);
                total = acc;
            }
            const int tc = (tj + dr) * dsz + (ti + dr);
            out[(((i64)b * oC + tc) * oH + oy) * oW + ox] = total / (float)nelems;
        }
    }
    return 0;
}

int vfi_oracle_correlation_bwd(const float* f1, const float* f2, const float* gout,
                               float* g1, float* g2, int B, int C, int H, int W,
                               int pad, int k, int md, int s1, int s2) {
    int oC, oH, oW;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
    if (vfi_oracle_correlation_out_dims(H, W, pad, k, md, s1, s2, &oC, &oH, &oW)) return 1;
    /* the reference's backward indexes gradInput rows by blockIdx*stride1, which
     * leaves the tensor for stride1 > 1: only stride1 == 1 is defined */
    if (s1 != 1 || oH <= 0 || oW <= 0) return 1;
    const int kr = (k - 1) / 2;
    const int dr = md / s2;
    const int dsz = 2 * dr + 1;
    const float nelems = (float)(k * k * C);
    memset(g1, 0, sizeof(float) * (size_t)B * C * H * W);     /* cc:112-113 */
    memset(g2, 0, sizeof(float) * (size_t)B * C * H * W);
    for (int n = 0; n < B; ++n)
    for (int by = 0; by < H; ++by)
    for (int bx = 0; bx < W; ++bx)
    for (int c = 0; c < C; ++c) {
        const int y = by * s1 + pad, x = bx * s1 + pad;
        /* ---- gradInput1 (:162-239) */
        do {
            int xmin = (x - kr - md) / s1, ymin = (y - kr - md) / s1;
            int xmax = (x + kr - md) / s1, ymax = (y + kr - md) / s1;
            if (xmax < 0 || ymax < 0 || xmin >= oW || ymin >= oH) break;
            if (xmin > xmax || ymin > ymax) break;
            xmin = maxi(0, xmin); xmax = mini(oW - 1, xmax);
            ymin = maxi(0, ymin); ymax = mini(oH - 1, ymax);
            float part[32];
            for (int l = 0; l < 32; ++l) {
                float s = 0.0f;
                for (int tc = l; tc < oC; tc += 32) {
                    const int i2 = (tc % dsz - dr) * s2, j2 = (tc / dsz - dr) * s2;
                    const float val2 = padded(f2, C, H, W, pad, n, c, y + j2, x + i2);
                    for (int j = ymin; j <= ymax; ++j)
                    for (int i = xmin; i <= xmax; ++i)
                        s = fmaf(gout[(((i64)n * oC + tc) * oH + j) * oW + i], val2, s);
                }
                part[l] = s;
            }
            float r = 0.0f;
            for (int l = 0; l < 32; ++l) r += part[l];
            g1[(((i64)n * C + c) * H + (y - pad)) * W + (x - pad)] = r / nelems;
        } while (0);
        /* ---- gradInput2 (:255-332) */
        {
            float part[32];
            for (int l = 0; l < 32; ++l) {
                float s = 0.0f;
                for (int tc = l; tc < oC; tc += 32) {
                    const int i2 = (tc % dsz - dr) * s2, j2 = (tc / dsz - dr) * s2;
                    int xmin = (x - kr - md - i2) / s1, ymin = (y - kr - md - j2) / s1;
                    int xmax = (x + kr - md - i2) / s1, ymax = (y + kr - md - j2) / s1;
                    if (xmax < 0 || ymax < 0 || xmin >= oW || ymin >= oH) continue;
                    if (xmin > xmax || ymin > ymax) continue;
                    xmin = maxi(0, xmin); xmax = mini(oW - 1, xmax);
                    ymin = maxi(0, ymin); ymax = mini(oH - 1, ymax);
                    const float val1 = padded(f1, C, H, W, pad, n, c, y - j2, x - i2);
                    for (int j = ymin; j <= ymax; ++j)
                    for (int i = xmin; i <= xmax; ++i)
                        s = fmaf(gout[(((i64)n * oC + tc) * oH + j) * oW + i], val1, s);
                }
                part[l] = s;
            }
            float r = 0.0f;
            for (int l = 0; l < 32; ++l) r += part[l];
            g2[(((i64)n * C + c) * H + (y - pad)) * W + (x - pad)] = r / nelems;
        }
    }
    return 0;
}

/* ------------------------------------------------------------ glue */

/* ATen area_pixel_compute_source_index, align_corners=False, scale 1/4: 0.25*(dst+0.5)-0.5 clamped
 * at 0; second tap one further unless at the last input index (UpSampleBilinear2d) */
static inline void up4_tap(int dst, int in_size, int* i0, int* i1, float* l0, float* l1) {
    float src = 0.25f * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    *i0 = (int)src;
    *i1 = *i0 + (*i0 < in_size - 1 ? 1 : 0);
    *l1 = src - (float)*i0;
    *l0 = 1.0f - *l1;
}

int vfi_oracle_flow_upsample4(const float* in, float* out, int B, int C, int hq, int wq,
                              float m0, float m1, int fmad) {
    if (B <= 0 || C <= 0 || hq <= 0 || wq <= 0) return 1;
    const int H = 4 * hq, W = 4 * wq;
    for (int bc = 0; bc < B * C; ++bc) {
        const float* p = in + (i64)bc * hq * wq;
        float* o = out + (i64)bc * H * W;
        for (int y = 0; y < H; ++y) {
            int y0, y1; float ly0, ly1;
            up4_tap(y, hq, &y0, &y1, &ly0, &ly1);
            for (int x = 0; x < W; ++x) {
                int x0, x1; float lx0, lx1;
                up4_tap(x, wq, &x0, &x1, &lx0, &lx1);
                /* div_flow * temp * time_offset: two float32 products, left to right */
                const float p00 = (m0 * p[(i64)y0 * wq + x0]) * m1, p01 = (m0 * p[(i64)y0 * wq + x1]) * m1;
                const float p10 = (m0 * p[(i64)y1 * wq + x0]) * m1, p11 = (m0 * p[(i64)y1 * wq + x1]) * m1;
                /* h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11) */
                const float t0 = mac(lx1, p01, lx0 * p00, fmad);
                const float t1 = mac(lx1, p11, lx0 * p10, fmad);
                o[(i64)y * W + x] = mac(ly1, t1, ly0 * t0, fmad);
            }
        }
    }
    return 0;
}

int vfi_oracle_pwc_warp(const float* x, const float* flo, float* out, int B, int C, int H, int W,
                        int align_corners, int fmad) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 1;
    const i64 HW = (i64)H * W;
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < H; ++y)
    for (int xx = 0; xx < W; ++xx) {
        const i64 px = (i64)y * W + xx;
        /* vgrid = grid + flo; 2.0*v/max(W-1,1) - 1.0   (PWCNet.py:181-185) */
        const float vx = (float)xx + flo[((i64)b * 2 + 0) * HW + px];
        const float vy = (float)y + flo[((i64)b * 2 + 1) * HW + px];
        const float gx = 2.0f * vx / (float)maxi(W - 1, 1) - 1.0f;
        const float gy = 2.0f * vy / (float)maxi(H - 1, 1) - 1.0f;
        /* ATen grid_sampler_unnormalize */
        const float ix = align_corners ? ((gx + 1.0f) / 2.0f) * (float)(W - 1) : ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
        const float iy = align_corners ? ((gy + 1.0f) / 2.0f) * (float)(H - 1) : ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float wnw = (fx0 + 1.0f - ix) * (fy0 + 1.0f - iy), wne = (ix - fx0) * (fy0 + 1.0f - iy);
        const float wsw = (fx0 + 1.0f - ix) * (iy - fy0), wse = (ix - fx0) * (iy - fy0);
        const int finite = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
        const int x0 = finite ? (int)fx0 : -2, y0 = finite ? (int)fy0 : -2;
        const int inx0 = x0 >= 0 && x0 < W, inx1 = x0 + 1 >= 0 && x0 + 1 < W;
        const int iny0 = y0 >= 0 && y0 < H, iny1 = y0 + 1 >= 0 && y0 + 1 < H;
        /* grid_sample of the ones tensor, then mask[mask<0.9999]=0; mask[mask>0]=1 */
        float m = 0.0f;
        if (iny0 && inx0) m += wnw;
        if (iny0 && inx1) m += wne;
        if (iny1 && inx0) m += wsw;
        if (iny1 && inx1) m += wse;
        const float mask = (m < 0.9999f) ? 0.0f : (m > 0.0f ? 1.0f : m);
        for (int c = 0; c < C; ++c) {
            const float* p = x + ((i64)b * C + c) * HW;
            float v = 0.0f;
            if (iny0 && inx0) v = mac(p[(i64)y0 * W + x0], wnw, v, fmad);
            if (iny0 && inx1) v = mac(p[(i64)y0 * W + x0 + 1], wne, v, fmad);
            if (iny1 && inx0) v = mac(p[(i64)(y0 + 1) * W + x0], wsw, v, fmad);
            if (iny1 && inx1) v = mac(p[(i64)(y0 + 1) * W + x0 + 1], wse, v, fmad);
            out[((i64)b * C + c) * HW + px] = v * mask;
        }
    }
    return 0;
}
