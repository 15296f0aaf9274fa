// glue.hip -- the steps immediately either side of the hot-path ops, as single
// launches for gfx950:
//   * FilterInterpolate of both reference frames + the time-weighted blend
//     (networks/DAIN_slowmotion.py:324-335, networks/DAIN.py:560-573);
//   * PWC-Net's warp(): grid normalisation, grid_sample of the features and of a ones mask,
//     threshold, multiply (PWCNet/PWCNet.py:159-199);
//   * the frame boundary: uint8 HWC -> float32 CHW / 255 with replication padding, and back
//     (clip, crop, x255, round half to even, uint8), plus the sums behind PSNR and the
//     interpolation error (demo_MiddleBury.py:280-318, 350-364, 370-388).
// (The x4 flow upsample fused into the projection is in projection.hip.)
#include "filterinterp_dev.h"

namespace vfi {

// ------------------------------------------------------------------ FilterInterpolate x2 + blend

struct FiSide {
    bool valid;
    int L, T, ix, iy;
    float alpha, beta;
};
__device__ __forceinline__ FiSide fi_side(const float* __restrict__ flow, int64_t cstride, int x, int y, int w, int h, int fs) {
    FiSide s;
    const float fx = flow[0], fy = flow[cstride];
    const float x2 = (float)x + fx, y2 = (float)y + fy;
    s.valid = fi_valid(fx, fy, x2, y2, w, h);
    s.ix = s.valid ? (int)x2 : 0;
    s.iy = s.valid ? (int)y2 : 0;
    s.L = s.ix + 1 - fs / 2;
    s.T = s.iy + 1 - fs / 2;
    s.alpha = x2 - (float)s.ix;
    s.beta = y2 - (float)s.iy;
    return s;
}
__device__ __forceinline__ float fi_side_value(const FiSide& s, const float* __restrict__ plane, const float* __restrict__ fpx,
                                               int64_t fcs, int hs, int h, int w, int fs, int x, int y) {
    if (!s.valid) return plane[(int64_t)y * hs + x];       // copy-through (:2814-2818)
    float q[4];
    quadrants_generic(plane, fpx, fcs, hs, h, w, fs, s.L, s.T, s.ix, s.iy, q);
    return blend4(s.alpha, s.beta, q[0], q[1], q[2], q[3]);
}

// out0 = FI(ref0, flow0, filt0), out2 = FI(ref2, flow2, filt2), blend = out0 * w0 + out2 * w2 with the
// two products rounded separately (torch evaluates a*(1-t) + b*t as three elementwise ops)
__global__ __launch_bounds__(VFI_TX * VFI_TY) void fi_blend_forward(
    const float* __restrict__ ref0, const float* __restrict__ ref2, const float* __restrict__ flow0,
    const float* __restrict__ flow2, const float* __restrict__ filt0, const float* __restrict__ filt2,
    float* __restrict__ blend, float* __restrict__ out0, float* __restrict__ out2,
    int channel, int h, int w, int fs, float w0, float w2,
    vfi_strides sr, vfi_strides sf, vfi_strides sk, vfi_strides so) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const int64_t fo = (int64_t)b * sf.b + (int64_t)y * sf.h + x, ko = (int64_t)b * sk.b + (int64_t)y * sk.h + x;
    const FiSide a = fi_side(flow0 + fo, sf.c, x, y, w, h, fs), c2 = fi_side(flow2 + fo, sf.c, x, y, w, h, fs);
    const int64_t oo = (int64_t)b * so.b + (int64_t)y * so.h + x;
    if (fs == 4) {
        // taps, clamped rows and columns of both sides hoisted out of the channel loop
        float f0[16], f2[16];
        unsigned ro0[4], co0[4], ro2[4], co2[4];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f0[k] = a.valid ? filt0[ko + (int64_t)k * sk.c] : 0.0f;
            f2[k] = c2.valid ? filt2[ko + (int64_t)k * sk.c] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ro0[k] = (unsigned)(clampi(a.T + k, 0, h - 1) * (int)sr.h);  co0[k] = (unsigned)clampi(a.L + k, 0, w - 1);
            ro2[k] = (unsigned)(clampi(c2.T + k, 0, h - 1) * (int)sr.h); co2[k] = (unsigned)clampi(c2.L + k, 0, w - 1);
        }
        const unsigned self = (unsigned)(y * (int)sr.h + x);
        for (int c = 0; c < channel; ++c) {
            const float* p0 = ref0 + (int64_t)b * sr.b + (int64_t)c * sr.c;
            const float* p2 = ref2 + (int64_t)b * sr.b + (int64_t)c * sr.c;
            const float v0 = a.valid ? fi4_value(p0, ro0, co0, f0, a.alpha, a.beta) : p0[self];
            const float v2 = c2.valid ? fi4_value(p2, ro2, co2, f2, c2.alpha, c2.beta) : p2[self];
            if (out0) out0[oo + (int64_t)c * so.c] = v0;
            if (out2) out2[oo + (int64_t)c * so.c] = v2;
            const float q0 = v0 * w0, q2 = v2 * w2;
            blend[oo + (int64_t)c * so.c] = q0 + q2;
        }
        return;
    }
    for (int c = 0; c < channel; ++c) {
        const float v0 = fi_side_value(a, ref0 + (int64_t)b * sr.b + (int64_t)c * sr.c, filt0 + ko, sk.c, (int)sr.h, h, w, fs, x, y);
        const float v2 = fi_side_value(c2, ref2 + (int64_t)b * sr.b + (int64_t)c * sr.c, filt2 + ko, sk.c, (int)sr.h, h, w, fs, x, y);
        if (out0) out0[oo + (int64_t)c * so.c] = v0;
        if (out2) out2[oo + (int64_t)c * so.c] = v2;
        const float p0 = v0 * w0, p2 = v2 * w2;
        blend[oo + (int64_t)c * so.c] = p0 + p2;
    }
}

// blend = out0 * w0 + out2 * w2, the two products rounded separately
__global__ __launch_bounds__(VFI_TX * VFI_TY) void fi_blend_only(
    const float* __restrict__ out0, const float* __restrict__ out2, float* __restrict__ blend, int channel, int h, int w,
    float w0, float w2, vfi_strides so) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int64_t oo = (int64_t)blockIdx.z * so.b + (int64_t)y * so.h + x;
    for (int c = 0; c < channel; ++c) {
        const float q0 = out0[oo + (int64_t)c * so.c] * w0, q2 = out2[oo + (int64_t)c * so.c] * w2;
        blend[oo + (int64_t)c * so.c] = q0 + q2;
    }
}

// ------------------------------------------------------------------ PWC-Net warp()

// vgrid = pixel + flow; normalised as PWCNet.py:184-185; grid_sample (bilinear, zeros padding) of x
// and of a ones tensor; mask = (ones sample >= 0.9999); out = sample * mask.  align_corners selects
// grid_sample's un-normalisation: 1 = torch <= 1.2 (what the reference was written for: the
// normalisation above then round-trips to pixel + flow), 0 = the default of torch >= 1.3.
#define PWC_CH 8                    // channels per thread: blockIdx.z = batch x channel chunks
__global__ __launch_bounds__(VFI_TX * VFI_TY) void pwc_warp_forward(
    const float* __restrict__ xin, const float* __restrict__ flo, float* __restrict__ out,
    int channel, int groups, int h, int w, int align_corners, vfi_strides sx, vfi_strides sf, vfi_strides so) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z / groups;
    const float* f = flo + (int64_t)b * sf.b + (int64_t)y * sf.h + x;
    const float vx = (float)x + f[0], vy = (float)y + f[sf.c];
    const float gx = 2.0f * vx / (float)max(w - 1, 1) - 1.0f;
    const float gy = 2.0f * vy / (float)max(h - 1, 1) - 1.0f;
    // ATen grid_sampler_unnormalize
    const float ix = align_corners ? ((gx + 1.0f) / 2.0f) * (float)(w - 1) : ((gx + 1.0f) * (float)w - 1.0f) / 2.0f;
    const float iy = align_corners ? ((gy + 1.0f) / 2.0f) * (float)(h - 1) : ((gy + 1.0f) * (float)h - 1.0f) / 2.0f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    // corners as ATen orders them: nw, ne, sw, se; weights from the opposite corner
    const float wnw = (fx0 + 1.0f - ix) * (fy0 + 1.0f - iy), wne = (ix - fx0) * (fy0 + 1.0f - iy);
    const float wsw = (fx0 + 1.0f - ix) * (iy - fy0), wse = (ix - fx0) * (iy - fy0);
    // float -> int of a huge or NaN coordinate is undefined in C; such a corner is out of bounds anyway
    const bool finite = fabsf(ix) < 1.0e9f && fabsf(iy) < 1.0e9f;
    const int x0 = finite ? (int)fx0 : -2, y0 = finite ? (int)fy0 : -2;
    const bool inx0 = x0 >= 0 && x0 < w, inx1 = x0 + 1 >= 0 && x0 + 1 < w;
    const bool iny0 = y0 >= 0 && y0 < h, iny1 = y0 + 1 >= 0 && y0 + 1 < h;
    float m = 0.0f;
    if (iny0 && inx0) m += wnw;
    if (iny0 && inx1) m += wne;
    if (iny1 && inx0) m += wsw;
    if (iny1 && inx1) m += wse;
    const float mask = (m < 0.9999f) ? 0.0f : (m > 0.0f ? 1.0f : m);   // mask[mask<0.9999]=0; mask[mask>0]=1 (NaN stays)
    // A corner outside the map is skipped by ATen; here it is read at a clamped (valid) address and both the value
    // and its weight are replaced by 0, so v = fma(0, 0, v) = v: the same result from straight-line loads, all of a
    // chunk's 4 x PWC_CH loads in flight at once.  (With the conditional loads of a per-channel loop a thread walked
    // its channels one memory round trip at a time: the 196-channel 18x31 level took longer than the 32-channel
    // 288x496 one.)
    const int cx0 = clampi(x0, 0, w - 1), cx1 = clampi(x0 + 1, 0, w - 1), cy0 = clampi(y0, 0, h - 1), cy1 = clampi(y0 + 1, 0, h - 1);
    const int64_t onw = (int64_t)cy0 * sx.h + cx0, one = (int64_t)cy0 * sx.h + cx1;
    const int64_t osw = (int64_t)cy1 * sx.h + cx0, ose = (int64_t)cy1 * sx.h + cx1;
    const bool bnw = iny0 && inx0, bne = iny0 && inx1, bsw = iny1 && inx0, bse = iny1 && inx1;
    const float enw = bnw ? wnw : 0.0f, ene = bne ? wne : 0.0f, esw = bsw ? wsw : 0.0f, ese = bse ? wse : 0.0f;
    const int c0 = (int)(blockIdx.z % groups) * PWC_CH;
    const float* src = xin + (int64_t)b * sx.b + (int64_t)c0 * sx.c;
    float* dst = out + (int64_t)b * so.b + (int64_t)c0 * so.c + (int64_t)y * so.h + x;
    auto one_channel = [&](float pnw, float pne, float psw, float pse) {
        float v = 0.0f;                                     // out_acc += value * weight, fused as nvcc fuses ATen's grid_sampler
        v = fmaf(bnw ? pnw : 0.0f, enw, v);
        v = fmaf(bne ? pne : 0.0f, ene, v);
        v = fmaf(bsw ? psw : 0.0f, esw, v);
        v = fmaf(bse ? pse : 0.0f, ese, v);
        return v * mask;
    };
    if (c0 + PWC_CH <= channel) {
        float q[PWC_CH][4];
#pragma unroll
        for (int c = 0; c < PWC_CH; ++c) {
            const float* pl = src + (int64_t)c * sx.c;
            q[c][0] = pl[onw]; q[c][1] = pl[one]; q[c][2] = pl[osw]; q[c][3] = pl[ose];
        }
#pragma unroll
        for (int c = 0; c < PWC_CH; ++c) dst[(int64_t)c * so.c] = one_channel(q[c][0], q[c][1], q[c][2], q[c][3]);
    } else {
        for (int c = 0; c0 + c < channel; ++c) {
            const float* pl = src + (int64_t)c * sx.c;
            dst[(int64_t)c * so.c] = one_channel(pl[onw], pl[one], pl[osw], pl[ose]);
        }
    }
}

// ------------------------------------------------------------------ frame boundary

// dst[c][y][x] = src[clamp(y - top)][clamp(x - left)][c] / 255  (astype(float32) / 255.0, ReplicationPad2d)
__global__ __launch_bounds__(VFI_TX * VFI_TY) void frame_u8_to_planar(
    const unsigned char* __restrict__ src, float* __restrict__ dst, int h, int w, int top, int left, int ph, int pw,
    vfi_strides sd) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= pw || y >= ph) return;
    const int b = blockIdx.z;
    const unsigned char* p = src + ((int64_t)b * h * w + (int64_t)clampi(y - top, 0, h - 1) * w + clampi(x - left, 0, w - 1)) * 3;
    float* d = dst + (int64_t)b * sd.b + (int64_t)y * sd.h + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) d[(int64_t)c * sd.c] = (float)p[c] / 255.0f;
}

// dst[y][x][c] = uint8(rint(255 * clip(src[c][top + y][left + x], 0, 1)))   (np.round: half to even)
__global__ __launch_bounds__(VFI_TX * VFI_TY) void planar_to_frame_u8(
    const float* __restrict__ src, unsigned char* __restrict__ dst, int h, int w, int top, int left, vfi_strides ss) {
    const int x = blockIdx.x * VFI_TX + threadIdx.x;
    const int y = blockIdx.y * VFI_TY + threadIdx.y;
    if (x >= w || y >= h) return;
    const int b = blockIdx.z;
    const float* p = src + (int64_t)b * ss.b + (int64_t)(top + y) * ss.h + left + x;
    unsigned char* d = dst + ((int64_t)b * h * w + (int64_t)y * w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = p[(int64_t)c * ss.c];
        v = fminf(fmaxf(v, 0.0f), 1.0f);                    // NaN -> 0 here; numpy's clip keeps NaN (then astype is undefined)
        d[c] = (unsigned char)(int)rintf(255.0f * v);
    }
}

// sums[0] += sum |a - b|, sums[1] += sum (a - b)^2 over n bytes: exact integers.  16 bytes per load,
// one pair of atomics per workgroup.
__global__ __launch_bounds__(256) void frame_error_sums(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                                        int64_t n, unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long part[2][4];
    unsigned long long sa = 0ull, sq = 0ull;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool wide = (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    const int64_t n16 = wide ? n >> 4 : 0;
    for (int64_t i = tid; i < n16; i += stride) {
        const uint4 va = reinterpret_cast<const uint4*>(a)[i], vb = reinterpret_cast<const uint4*>(b)[i];
        const unsigned wa[4] = { va.x, va.y, va.z, va.w }, wb[4] = { vb.x, vb.y, vb.z, vb.w };
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int s = 0; s < 32; s += 8) {
                const int d = (int)((wa[k] >> s) & 255u) - (int)((wb[k] >> s) & 255u);
                sa += (unsigned)(d < 0 ? -d : d);
                sq += (unsigned)(d * d);
            }
    }
    for (int64_t i = (n16 << 4) + tid; i < n; i += stride) {
        const int d = (int)a[i] - (int)b[i];
        sa += (unsigned)(d < 0 ? -d : d);
        sq += (unsigned)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sa += __shfl_xor(sa, o);
        sq += __shfl_xor(sq, o);
    }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = sa; part[1][threadIdx.x >> 6] = sq; }
    __syncthreads();
    if (threadIdx.x < 2)
        atomicAdd(&sums[threadIdx.x], part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3]);
}

// ---- SSIM of two uint8 frames as demo_MiddleBury.py:382-388 calls it: each colour plane scaled to [0,1] is one
// single-channel image, 11-tap Gaussian (sigma 1.5) along H then along W without padding (a dimension shorter
// than the window is left unsmoothed, :112-119), data_range 1, K = (0.01, 0.03) (:125-162).
// One workgroup = a 32x16 tile of SSIM values of one plane: the (16+10)x(32+10) pixels of both frames go to LDS,
// the vertical pass leaves five blurred fields (x, y, xx, yy, xy) there, the horizontal pass finishes two
// values per thread.  Every value is added as a 2^-32 fixed-point integer: the total does not depend on the
// order of the atomics.
#define SSIM_TX 32
#define SSIM_TY 16
#define SSIM_TAPS 11
struct SsimWindow { float wy[SSIM_TAPS], wx[SSIM_TAPS]; int ny, nx; };

__global__ __launch_bounds__(256) void frame_ssim_sums(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                                       int h, int w, SsimWindow win, long long* __restrict__ sums) {
    constexpr int IW = SSIM_TX + SSIM_TAPS - 1, IH = SSIM_TY + SSIM_TAPS - 1;
    __shared__ float px[2][IH][IW];
    __shared__ float mid[5][SSIM_TY][IW + 1];
    __shared__ long long part[4];
    const int oh = h - win.ny + 1, ow = w - win.nx + 1;     // valid outputs
    const int plane = blockIdx.z % 3, frame = blockIdx.z / 3;
    const int x0 = blockIdx.x * SSIM_TX, y0 = blockIdx.y * SSIM_TY;
    const int64_t base = (int64_t)frame * h * w * 3 + plane;
    const int rows = min(SSIM_TY, oh - y0) + win.ny - 1, cols = min(SSIM_TX, ow - x0) + win.nx - 1;
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
        const int r = i / IW, c = i % IW;
        float va = 0.0f, vb = 0.0f;
        if (r < rows && c < cols) {
            const int64_t o = base + ((int64_t)(y0 + r) * w + (x0 + c)) * 3;
            va = (float)a[o] / 255.0f;                      // ToTensor: uint8 -> float32 / 255
            vb = (float)b[o] / 255.0f;
        }
        px[0][r][c] = va;
        px[1][r][c] = vb;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SSIM_TY * IW; i += 256) {
        const int r = i / IW, c = i % IW;
        float sx = 0.0f, sy = 0.0f, sxx = 0.0f, syy = 0.0f, sxy = 0.0f;
        for (int k = 0; k < win.ny; ++k) {
            const float g = win.wy[k], u = px[0][r + k][c], v = px[1][r + k][c];
            sx = fmaf(g, u, sx);
            sy = fmaf(g, v, sy);
            sxx = fmaf(g, u * u, sxx);
            syy = fmaf(g, v * v, syy);
            sxy = fmaf(g, u * v, sxy);
        }
        mid[0][r][c] = sx; mid[1][r][c] = sy; mid[2][r][c] = sxx; mid[3][r][c] = syy; mid[4][r][c] = sxy;
    }
    __syncthreads();
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    long long acc = 0;
    for (int i = threadIdx.x; i < SSIM_TY * SSIM_TX; i += 256) {
        const int r = i / SSIM_TX, c = i % SSIM_TX;
        if (y0 + r >= oh || x0 + c >= ow) continue;
        float m[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            float t = 0.0f;
            for (int k = 0; k < win.nx; ++k) t = fmaf(win.wx[k], mid[q][r][c + k], t);
            m[q] = t;
        }
        const float mu1_sq = m[0] * m[0], mu2_sq = m[1] * m[1], mu12 = m[0] * m[1];
        const float s1 = m[2] - mu1_sq, s2 = m[3] - mu2_sq, s12 = m[4] - mu12;
        const float cs = (2.0f * s12 + C2) / (s1 + s2 + C2);
        const float v = ((2.0f * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs;
        acc += __double2ll_rn((double)v * 4294967296.0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(reinterpret_cast<unsigned long long*>(sums), (unsigned long long)(part[0] + part[1] + part[2] + part[3]));
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_filterinterp_forward_ori_lds_blend(const float* input1, const float* input2, const float* input3,
                                                       float* output, const float* other, float* blend, float w0, float w2,
                                                       int batch, int channel, int h, int w,
                                                       vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_stream_t stream);

extern "C" int vfi_filterinterp_blend_forward(const float* ref0, const float* ref2, const float* flow0, const float* flow2,
                                               const float* filt0, const float* filt2, float* blend, float* out0,
                                               float* out2, int batch, int channel, int h, int w, int filter_channels,
                                               float w0, float w2, vfi_strides s_ref, vfi_strides s_flow,
                                               vfi_strides s_filt, vfi_strides s_out, vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || filter_channels <= 0) return VFI_ERR_SHAPE;
    if (!ref0 || !ref2 || !flow0 || !flow2 || !filt0 || !filt2 || !blend) return VFI_ERR_SHAPE;
    const int fs = (int)sqrtf((float)filter_channels);
    if (filter_channels == 16 && out0 && out2 && s_out.b == s_ref.b && s_out.c == s_ref.c && s_out.h == s_ref.h) {
        // the LDS-staged forward per side (faster than the direct gather even at C = 3; it writes with the
        // input's strides), then the blend
        int err = vfi_filterinterp_forward_ori(ref0, flow0, filt0, out0, batch, channel, h, w, 16, s_ref, s_flow, s_filt, stream);
        if (err != VFI_OK) return err;
        // the blend as the second launch's epilogue (3-channel frames); else a launch of its own
        err = vfi_filterinterp_forward_ori_lds_blend(ref2, flow2, filt2, out2, out0, blend, w0, w2, batch, channel, h, w, s_ref, s_flow,
                                                     s_filt, stream);
        if (err != -1) return err;
        err = vfi_filterinterp_forward_ori(ref2, flow2, filt2, out2, batch, channel, h, w, 16, s_ref, s_flow, s_filt, stream);
        if (err != VFI_OK) return err;
        hipLaunchKernelGGL(fi_blend_only, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream, out0,
                           out2, blend, channel, h, w, w0, w2, s_out);
        return launch_status();
    }
    hipLaunchKernelGGL(fi_blend_forward, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream, ref0,
                       ref2, flow0, flow2, filt0, filt2, blend, out0, out2, channel, h, w, fs, w0, w2, s_ref, s_flow,
                       s_filt, s_out);
    return launch_status();
}

extern "C" int vfi_pwc_warp_forward(const float* x, const float* flow, float* output, int batch, int channel, int h, int w,
                                     int align_corners, vfi_strides sx, vfi_strides sf, vfi_strides so,
                                     vfi_stream_t stream) {
    if (batch <= 0 || channel <= 0 || h <= 0 || w <= 0 || !x || !flow || !output) return VFI_ERR_SHAPE;
    const int groups = (channel + PWC_CH - 1) / PWC_CH;
    if ((int64_t)batch * groups > 65535) return VFI_ERR_SHAPE;
    dim3 grid = pixel_grid(w, h, batch);
    grid.z = (unsigned)(batch * groups);
    hipLaunchKernelGGL(pwc_warp_forward, grid, dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream, x, flow,
                       output, channel, groups, h, w, align_corners ? 1 : 0, sx, sf, so);
    return launch_status();
}

extern "C" int vfi_frame_u8_to_planar(const unsigned char* src_hwc, float* dst, int batch, int h, int w, int pad_left,
                                       int pad_right, int pad_top, int pad_bottom, vfi_strides sd, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || pad_left < 0 || pad_right < 0 || pad_top < 0 || pad_bottom < 0 || !src_hwc || !dst)
        return VFI_ERR_SHAPE;
    const int ph = h + pad_top + pad_bottom, pw = w + pad_left + pad_right;
    hipLaunchKernelGGL(frame_u8_to_planar, pixel_grid(pw, ph, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream,
                       src_hwc, dst, h, w, pad_top, pad_left, ph, pw, sd);
    return launch_status();
}

extern "C" int vfi_planar_to_frame_u8(const float* src, unsigned char* dst_hwc, int batch, int h, int w, int top, int left,
                                       vfi_strides ss, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || top < 0 || left < 0 || !src || !dst_hwc) return VFI_ERR_SHAPE;
    hipLaunchKernelGGL(planar_to_frame_u8, pixel_grid(w, h, batch), dim3(VFI_TX, VFI_TY, 1), 0, (hipStream_t)stream, src,
                       dst_hwc, h, w, top, left, ss);
    return launch_status();
}

extern "C" int vfi_frame_error_sums(const unsigned char* a, const unsigned char* b, int64_t n, unsigned long long* sums,
                                     vfi_stream_t stream) {
    if (n <= 0 || !a || !b || !sums) return VFI_ERR_SHAPE;
    const int64_t blocks = (n + 256 * 64 - 1) / (256 * 64);
    hipLaunchKernelGGL(frame_error_sums, dim3((unsigned)(blocks < 512 ? blocks : 512)), dim3(256), 0, (hipStream_t)stream,
                       a, b, n, sums);
    return launch_status();
}

extern "C" int vfi_frame_ssim_sums(const unsigned char* a, const unsigned char* b, int batch, int h, int w,
                                    long long* sums, vfi_stream_t stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || !a || !b || !sums) return VFI_ERR_SHAPE;
    SsimWindow win;
    double g[SSIM_TAPS], total = 0.0;
    for (int k = 0; k < SSIM_TAPS; ++k) {                  // exp(-x^2 / (2 sigma^2)), normalised (:98-109)
        const double x = (double)(k - SSIM_TAPS / 2);
        g[k] = exp(-(x * x) / (2.0 * 1.5 * 1.5));
        total += g[k];
    }
    for (int k = 0; k < SSIM_TAPS; ++k) win.wy[k] = win.wx[k] = (float)(g[k] / total);
    win.ny = win.nx = SSIM_TAPS;
    if (h < SSIM_TAPS) { win.ny = 1; win.wy[0] = 1.0f; }   // a short dimension is not smoothed (:112-119)
    if (w < SSIM_TAPS) { win.nx = 1; win.wx[0] = 1.0f; }
    const int oh = h - win.ny + 1, ow = w - win.nx + 1;
    if ((int64_t)batch * 3 > 65535) return VFI_ERR_SHAPE;
    const dim3 grid((unsigned)((ow + SSIM_TX - 1) / SSIM_TX), (unsigned)((oh + SSIM_TY - 1) / SSIM_TY), (unsigned)(batch * 3));
    hipLaunchKernelGGL(frame_ssim_sums, grid, dim3(256), 0, (hipStream_t)stream, a, b, h, w, win, sums);
    return launch_status();
}
