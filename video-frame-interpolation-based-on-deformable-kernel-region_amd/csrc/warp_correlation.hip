// warp_correlation.hip -- PWC-Net's warp feeding its correlation layer, in one launch.
//
// PWCDCNet.forward (PWCNet/PWCNet.py:244-247, 266-267, 282-283, 299-300) does, at four of its five pyramid levels,
//     warp_k = self.warp(c2_k, up_flow * s)           # grid_sample + mask, PWCNet.py:159-199
//     corr_k = self.corr(c1_k, warp_k)                # correlation_cuda.forward, pad 4, k 1, md 4, strides 1
// and uses warp_k for nothing else.  Here the correlation kernel's staging loads of the second map ARE the warp: a
// window element (tile + 4 halo) is the bilinear sample of c2 at (x + flow) times the validity mask, formed exactly as
// vfi_pwc_warp_forward forms it (glue.hip) and written to LDS instead of HBM; the 81 products per pixel and channel are
// then those of vfi_correlation_forward, same sequential channel order.  The warped tensor never exists: per level one
// launch instead of two, and its write + read (2 x 18 MB at the finest 1080p level) are gone.  Results equal
// vfi_pwc_warp_forward followed by vfi_correlation_forward bit for bit.
//
// One kernel for every level size: 32 x 4 output pixels per workgroup, one wave per displacement row (9 waves), a
// lane owns two adjacent pixels; window 12 x 40 = 480 pixels, one per thread: a thread keeps its window pixel's
// four clamped tap offsets, four weights and mask in registers for all channels and fetches 8 channels x 4 taps
// per chunk, the next chunk's before the current one is multiplied.
#include "vfi_common.h"

namespace vfi {

#define WC_CC 8                     // channels per LDS fill
#define WC_MD 4
#define WC_D (2 * WC_MD + 1)
#define WC_TW 32
#define WC_TH 4
#define WC_LW (WC_TW + 2 * WC_MD)   // 40
#define WC_LH (WC_TH + 2 * WC_MD)   // 12
#define WC_NT (64 * WC_D)           // 576 threads
#define WC_F1 (WC_CC * WC_TH * WC_TW)               // first-map values per chunk: 1024
#define WC_NF1 ((WC_F1 + WC_NT - 1) / WC_NT)        // 2 per thread

__global__ __launch_bounds__(WC_NT) void warp_corr_forward(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ flo, float* __restrict__ out,
    int channel, int h, int w, int align_corners, vfi_strides sf) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) float tile[WC_CC][WC_LH][WC_LW];
    __shared__ __attribute__((aligned(16))) float f1s[WC_CC][WC_TH * WC_TW];

    const int lane = threadIdx.x, tj = threadIdx.y;
    const int tid = tj * 64 + lane;
    const int px = 2 * (lane & 15), py = lane >> 4;
    const int ox = blockIdx.x * WC_TW + px, oy = blockIdx.y * WC_TH + py;
    const int b = blockIdx.z;
    const int64_t plane = (int64_t)h * w;
    const float* f1 = in1 + (int64_t)b * channel * plane;
    const float* f2 = in2 + (int64_t)b * channel * plane;
    const int wy0 = blockIdx.y * WC_TH - WC_MD, wx0 = blockIdx.x * WC_TW - WC_MD;   // window origin (pad == md: org = 0)

    // ---- this thread's window pixel: the warp's sampling geometry (glue.hip: pwc_warp_forward, statement for statement)
    const int wr = tid / WC_LW, wc = tid - wr * WC_LW;
    const int gy = wy0 + wr, gx = wx0 + wc;
    const bool wpix = tid < WC_LH * WC_LW;
    const bool inframe = wpix && gy >= 0 && gy < h && gx >= 0 && gx < w;        // else the correlation's zero padding
    int onw = 0, one = 0, osw = 0, ose = 0;
    float enw = 0.0f, ene = 0.0f, esw = 0.0f, ese = 0.0f, mask = 0.0f;
    bool bnw = false, bne = false, bsw = false, bse = false;
    if (inframe) {
        const float* f = flo + (int64_t)b * sf.b + (int64_t)gy * sf.h + gx;
        const float vx = (float)gx + f[0], vy = (float)gy + f[sf.c];
        const float nx = 2.0f * vx / (float)max(w - 1, 1) - 1.0f;
        const float ny = 2.0f * vy / (float)max(h - 1, 1) - 1.0f;
        const float ix = align_corners ? ((nx + 1.0f) / 2.0f) * (float)(w - 1) : ((nx + 1.0f) * (float)w - 1.0f) / 2.0f;
        const float iy = align_corners ? ((ny + 1.0f) / 2.0f) * (float)(h - 1) : ((ny + 1.0f) * (float)h - 1.0f) / 2.0f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float wnw = (fx0 + 1.0f - ix) * (fy0 + 1.0f - iy), wne = (ix - fx0) * (fy0 + 1.0f - iy);
        const float wsw = (fx0 + 1.0f - ix) * (iy - fy0), wse = (ix - fx0) * (iy - fy0);
        const bool finite = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
        const int x0 = finite ? (int)fx0 : -2, y0 = finite ? (int)fy0 : -2;
        const bool inx0 = x0 >= 0 && x0 < w, inx1 = x0 + 1 >= 0 && x0 + 1 < w;
        const bool iny0 = y0 >= 0 && y0 < h, iny1 = y0 + 1 >= 0 && y0 + 1 < h;
        float m = 0.0f;
        if (iny0 && inx0) m += wnw;
        if (iny0 && inx1) m += wne;
        if (iny1 && inx0) m += wsw;
        if (iny1 && inx1) m += wse;
        mask = (m < 0.9999f) ? 0.0f : (m > 0.0f ? 1.0f : m);
        const int cx0 = clampi(x0, 0, w - 1), cx1 = clampi(x0 + 1, 0, w - 1), cy0 = clampi(y0, 0, h - 1), cy1 = clampi(y0 + 1, 0, h - 1);
        onw = cy0 * w + cx0; one = cy0 * w + cx1; osw = cy1 * w + cx0; ose = cy1 * w + cx1;
        bnw = iny0 && inx0; bne = iny0 && inx1; bsw = iny1 && inx0; bse = iny1 && inx1;
        enw = bnw ? wnw : 0.0f; ene = bne ? wne : 0.0f; esw = bsw ? wsw : 0.0f; ese = bse ? wse : 0.0f;
    }
    auto warped = [&](float pnw, float pne, float psw, float pse) {
        float v = 0.0f;
        v = fmaf(bnw ? pnw : 0.0f, enw, v);
        v = fmaf(bne ? pne : 0.0f, ene, v);
        v = fmaf(bsw ? psw : 0.0f, esw, v);
        v = fmaf(bse ? pse : 0.0f, ese, v);
        return v * mask;
    };

    // ---- staging plan of the first map: value e = tid + k * NT of the chunk's [CC][TH][TW] block
    int foff[WC_NF1], fch[WC_NF1];
    bool fok[WC_NF1];
#pragma unroll
    for (int k = 0; k < WC_NF1; ++k) {
        const int e = tid + k * WC_NT;
        const int c = e / (WC_TH * WC_TW), rem = e - c * (WC_TH * WC_TW);
        const int y = blockIdx.y * WC_TH + rem / WC_TW, x = blockIdx.x * WC_TW + rem % WC_TW;
        fch[k] = c;
        fok[k] = e < WC_F1 && y < h && x < w;
        foff[k] = fok[k] ? y * w + x : 0;
    }

    float acc[2][WC_D];
#pragma unroll
    for (int ti = 0; ti < WC_D; ++ti) { acc[0][ti] = 0.0f; acc[1][ti] = 0.0f; }

    float q[WC_CC][4], nf[WC_NF1];
    auto fetch = [&](int c0) {
        const int cn = min(WC_CC, channel - c0);
#pragma unroll
        for (int c = 0; c < WC_CC; ++c) {
            // (a channel past the end re-reads the last one: its values are never used)
            const float* pl = f2 + (int64_t)(c0 + min(c, cn - 1)) * plane;
            q[c][0] = pl[onw]; q[c][1] = pl[one]; q[c][2] = pl[osw]; q[c][3] = pl[ose];
        }
#pragma unroll
        for (int k = 0; k < WC_NF1; ++k) nf[k] = (fok[k] && fch[k] < cn) ? f1[(int64_t)(c0 + fch[k]) * plane + foff[k]] : 0.0f;
    };
    fetch(0);
    for (int c0 = 0; c0 < channel; c0 += WC_CC) {
        const int cn = min(WC_CC, channel - c0);
        __syncthreads();
        if (wpix) {
#pragma unroll
            for (int c = 0; c < WC_CC; ++c)
                tile[c][wr][wc] = (inframe && c < cn) ? warped(q[c][0], q[c][1], q[c][2], q[c][3]) : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < WC_NF1; ++k) {
            const int e = tid + k * WC_NT;
            if (e < WC_F1) (&f1s[0][0])[e] = nf[k];
        }
        __syncthreads();
        if (c0 + WC_CC < channel) fetch(c0 + WC_CC);
        for (int c = 0; c < cn; ++c) {
            const v2f a = *reinterpret_cast<const v2f*>(&f1s[c][py * WC_TW + px]);
            const v2f* row = reinterpret_cast<const v2f*>(&tile[c][py + tj][px]);
            float t[WC_D + 1];
#pragma unroll
            for (int k = 0; k < (WC_D + 1) / 2; ++k) {
                const v2f v = row[k];
                t[2 * k] = v.x;
                t[2 * k + 1] = v.y;
            }
#pragma unroll
            for (int ti = 0; ti < WC_D; ++ti) {
                acc[0][ti] = fmaf(a.x, t[ti], acc[0][ti]);
                acc[1][ti] = fmaf(a.y, t[ti + 1], acc[1][ti]);
            }
        }
    }
    const float nelems = (float)channel;
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (ox + k < w && oy < h) {
            float* o = out + ((int64_t)b * (WC_D * WC_D) + tj * WC_D) * plane + (int64_t)oy * w + ox + k;
#pragma unroll
            for (int ti = 0; ti < WC_D; ++ti) o[(int64_t)ti * plane] = acc[k][ti] / nelems;
        }
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_pwc_warp_correlation_forward(const float* input1, const float* input2, const float* flow, float* output,
                                                 int batch, int channel, int h, int w, int align_corners,
                                                 vfi_strides sf, vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || !input1 || !input2 || !flow || !output) return VFI_ERR_SHAPE;
    if ((int64_t)h * w > (1 << 30) || batch > 65535) return VFI_ERR_SHAPE;         // in-plane offsets are 32-bit
    const dim3 grid((w + WC_TW - 1) / WC_TW, (h + WC_TH - 1) / WC_TH, batch);
    hipLaunchKernelGGL(warp_corr_forward, grid, dim3(64, WC_D, 1), 0, (hipStream_t)stream, input1, input2, flow, output,
                       channel, h, w, align_corners ? 1 : 0, sf);
    return launch_status();
}
