// filterinterp_defor_bwd_lds.hip -- LDS-staged backward of the deformable FilterInterpolation variants, fs == 4.
//
// Semantics: filterinterpolation_cuda_kernel.cu:430-1215 (4-input backward, VARIANT 0), :1500-1935 (deforconv, 1),
// :2195-2567 (nofilterwithdeforconv, 2); the same arithmetic, in the same order, as fi_backward_defor in filterinterp.hip
// -- what differs is where the image values come from and where the image-gradient addends go.
//
// Why: per pixel and channel the backward samples 16 taps bilinearly (64 image values at addresses that jitter from lane
// to lane), and the per-tap kernel gathers them from global memory: 1.0 ms per call at 1080p, C = 3, against the staged
// forward's 0.14 ms (measured; bound by the latency of the gathers at two waves per SIMD).  Here a workgroup owns 64x4
// pixels and works through the channels three at a time: the bounding box of every corner of every displaced tap of
// the block is staged for the three planes by LDS-DMA (borders replicated while staging, as in
// filterinterp_defor_lds.hip), the taps read their corners from LDS, and the image-gradient addends -- exact 64-bit
// integers, scattered to the clamped UNDISPLACED taps as in the reference -- are summed in a second LDS window and leave
// with one global atomic per non-zero cell (vfi_common.h: gradacc_*; fi_backward_ori4_lds in filterinterp.hip).  Per tap
// the corner index, the two fractions, the weight and the three gradient sums (filter, offset y, offset x) stay in
// registers across the channels, summed in the reference's order from the cells' starting values: the per-tap kernel's
// bits.  No counted vmcnt here: a chunk's windows are waited for together (two workgroups per CU overlap).
//
// A block whose windows do not fit, a block with a non-finite offset, and every block of a call with non-finite
// gradients or weights raises its flag and returns; fi_backward_defor<V, false, 4>, launched afterwards, does those.
#include "filterinterp_dev.h"

#include <limits.h>

namespace vfi {

#define DB_TW 64
#define DB_TH 4
#define DB_THREADS (DB_TW * DB_TH)                  // 256: one pixel per thread
#define DB_CH 3                                     // channels per pass
#define DB_WIN_FLOATS 3840                          // staged window of one plane, at most (15 x 256)
#define DB_CELLS 4096                               // 64-bit cells of the gradient window, all channels of a pass

typedef __attribute__((address_space(3))) void* db_lptr_t;

template <int VARIANT>
__global__ __launch_bounds__(DB_THREADS, 2) void fi_backward_defor_lds(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ in3,
    const float* __restrict__ in4, const float* __restrict__ gout, unsigned long long* __restrict__ acc,
    const int* __restrict__ hdr, int* __restrict__ tileflag, float* g2, float* g3, float* g4,
    int channel, int h, int w, vfi_strides s1, vfi_strides s2, vfi_strides s3, vfi_strides s4) {
    constexpr int FS = 4, NT = 16;
    // ONE LDS array (a second __shared__ object beside an LDS-DMA target makes hipcc drain vmcnt before LDS reads):
    // header (two bounding boxes), the gradient cells, the image windows of a pass
    __shared__ __attribute__((aligned(16))) unsigned long long lds64[8 + DB_CELLS + DB_CH * DB_WIN_FLOATS / 2];
    int* box = reinterpret_cast<int*>(lds64);
    unsigned long long* cells = lds64 + 8;
    float* wins = reinterpret_cast<float*>(lds64 + 8 + DB_CELLS);

    const int tid = threadIdx.y * DB_TW + threadIdx.x;
    const int x = blockIdx.x * DB_TW + threadIdx.x;
    const int y = blockIdx.y * DB_TH + threadIdx.y;
    const int b = blockIdx.z;
    const int tile = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const GradAccCtx gctx = gradacc_ctx(hdr);
    const bool inimg = x < w && y < h;
    float fx = 0.0f, fy = 0.0f;
    if (inimg) {
        const float* flow = in2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        fx = flow[0];
        fy = flow[s2.c];
    }
    const float x2 = (float)x + fx;
    const float y2 = (float)y + fy;
    const bool valid = inimg && fi_valid(fx, fy, x2, y2, w, h);
    const int ix = valid ? (int)x2 : 0, iy = valid ? (int)y2 : 0;
    const int L = ix + 1 - FS / 2, T = iy + 1 - FS / 2;
    const float alpha = x2 - (float)ix;
    const float beta = y2 - (float)iy;

    // VARIANT 2: the third input IS the offset field and g3 its gradient; no filter
    const float* fpx = in3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    float* gfpx = g3 + (int64_t)b * s3.b + (int64_t)y * s3.h + x;
    const float* opx = (VARIANT == VFI_DEFOR_NOFILTER) ? fpx : in4 + (int64_t)b * s4.b + (int64_t)y * s4.h + x;
    float* gopx = (VARIANT == VFI_DEFOR_NOFILTER) ? gfpx : g4 + (int64_t)b * s4.b + (int64_t)y * s4.h + x;
    const int64_t ocs = (VARIANT == VFI_DEFOR_NOFILTER) ? s3.c : s4.c;

    // ---- the displaced taps of this pixel: corner (frame coordinates from -1: rows <= -1 all replicate row 0, so a corner
    // pair starting at clamp(Top, -1, h - 1) reads what clamp(Top), clamp(Top + 1) read), fractions, quadrant, weight
    int tcy[NT], tcx[NT];
    float phy[NT], phx[NT], wgt[NT];
    unsigned qx = 0u, qy = 0u;
    int ro[FS], co[FS];                                     // the clamped UNDISPLACED taps: where the image gradient goes
#pragma unroll
    for (int k = 0; k < FS; ++k) { ro[k] = clampi(T + k, 0, h - 1); co[k] = clampi(L + k, 0, w - 1); }
    int bx_lo = INT_MAX, by_lo = INT_MAX, bx_hi = INT_MIN, by_hi = INT_MIN;
    bool finite = true;
    if (valid) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            phy[k] = opx[(int64_t)k * ocs];
            phx[k] = opx[(int64_t)(NT + k) * ocs];
            wgt[k] = (VARIANT == VFI_DEFOR_NOFILTER) ? 1.0f : fpx[(int64_t)k * s3.c];
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const float fracY = (float)ro[k / FS] + phy[k];
            const float fracX = (float)co[k % FS] + phx[k];
            finite = finite && fabsf(fracY) < 1e9f && fabsf(fracX) < 1e9f;       // (false for NaN; far positions take the per-tap kernel too)
            const int Top = (int)fracY, Left = (int)fracX;
            phy[k] = fracY - (float)Top;
            phx[k] = fracX - (float)Left;
            if (fracX > x2) qx |= 1u << k;
            if (fracY > y2) qy |= 1u << k;
            tcy[k] = clampi(Top, -1, h - 1);
            tcx[k] = clampi(Left, -1, w - 1);
            bx_lo = min(bx_lo, tcx[k]); by_lo = min(by_lo, tcy[k]);
            bx_hi = max(bx_hi, tcx[k] + 1); by_hi = max(by_hi, tcy[k] + 1);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) { phy[k] = 0.0f; phx[k] = 0.0f; wgt[k] = 0.0f; tcy[k] = 0; tcx[k] = 0; }
    }

    // ---- bounding boxes of the block: every corner of every tap (image windows), every undisplaced tap (gradient cells)
    if (tid < 8) box[tid] = (tid & 2) ? INT_MIN : INT_MAX;  // [0,1] = min x, y; [2,3] = max; [4..7] likewise for the cells
    if (tid == 8) box[8] = 0;                               // a pixel of the block has a non-finite tap position
    __syncthreads();
    {
        const int x0 = wave_min_i32(bx_lo), y0w = wave_min_i32(by_lo);
        const int x1 = wave_max_i32(bx_hi), y1 = wave_max_i32(by_hi);
        const int cx0 = wave_min_i32(valid ? co[0] : INT_MAX), cy0 = wave_min_i32(valid ? ro[0] : INT_MAX);
        const int cx1 = wave_max_i32(valid ? co[FS - 1] : INT_MIN), cy1 = wave_max_i32(valid ? ro[FS - 1] : INT_MIN);
        const bool bad = __builtin_amdgcn_ballot_w64(!finite) != 0ull;
        if ((tid & 63) == 0) {
            if (x0 != INT_MAX) {
                atomicMin(&box[0], x0); atomicMin(&box[1], y0w); atomicMax(&box[2], x1); atomicMax(&box[3], y1);
                atomicMin(&box[4], cx0); atomicMin(&box[5], cy0); atomicMax(&box[6], cx1); atomicMax(&box[7], cy1);
            }
            if (bad) atomicOr(&box[8], 1);
        }
    }
    __syncthreads();
    if (box[0] == INT_MAX) return;                          // (block-uniform: no pixel of the block has a gradient)
    const int bx0 = box[0], by0 = box[1], bw = box[2] - box[0] + 1, bh = box[3] - box[1] + 1;
    const int gx0 = box[4], gy0 = box[5], gw = box[6] - box[4] + 1, ncell = gw * (box[7] - box[5] + 1);
    const int pitch = (bw + 31) & ~31;                      // (a multiple of the 32 banks: fi_pitch_for's fp32 case)
    const int64_t n64 = (int64_t)pitch * bh;
    const int pc = min(DB_CH, DB_CELLS / max(ncell, 1));   // channels per pass: as many as the gradient cells allow
    if (gctx.nonfinite || box[8] || n64 > DB_WIN_FLOATS || pc == 0) {      // (block-uniform)
        if (tid == 0) tileflag[tile] = 1;                   // left to fi_backward_defor<VARIANT, false, 4>
        return;
    }
    const int n = (int)n64;
    int lb[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) lb[k] = (tcy[k] - by0) * pitch + (tcx[k] - bx0);
    int crow[FS];                                           // first cell of the taps' rows, less the cell box's first column
#pragma unroll
    for (int k = 0; k < FS; ++k) crow[k] = (ro[k] - gy0) * gw - gx0;

    // the pixel's gradient sums start from what the tensors hold (the reference adds into them)
    float gfa[NT], goy[NT], gox[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        goy[k] = valid ? gopx[(int64_t)k * ocs] : 0.0f;
        gox[k] = valid ? gopx[(int64_t)(NT + k) * ocs] : 0.0f;
        gfa[k] = (valid && VARIANT != VFI_DEFOR_NOFILTER) ? gfpx[(int64_t)k * s3.c] : 0.0f;
    }

    const float* img = in1 + (int64_t)b * s1.b;
    unsigned long long* gimg = acc + (int64_t)b * channel * h * w;       // dense [b][c][y][x] fixed-point sums
    const float* gpx = gout + (int64_t)b * s1.b + (int64_t)y * s1.h + x;
    const int hs = (int)s1.h;
    const int plane_bytes = 4 * ((h - 1) * hs + w);
    const float inv_pitch = 1.0f / (float)pitch;
    const int wave_first = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
    const float kq[4] = { (1.0f - alpha) * (1.0f - beta), alpha * (1.0f - beta), (1.0f - alpha) * beta, alpha * beta };
    float gx = 0.0f, gy = 0.0f;

    for (int c0 = 0; c0 < channel; c0 += pc) {
        const int cn = min(pc, channel - c0);
        // ---- stage the windows of the pass (element e = tid + k * 256 of a window, row-major with `pitch`; pad elements
        // and rows past the last get an out-of-range offset: zero, no memory traffic), zero the cells, fetch gradoutput
        for (int cc = 0; cc < cn; ++cc) {
            const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)(c0 + cc) * s1.c), 0, plane_bytes, 0x00020000);
            float* slot = wins + cc * DB_WIN_FLOATS + wave_first;
            for (int e0 = 0; e0 < n; e0 += DB_THREADS) {
                const int e = e0 + tid;
                const int r = fi_row_of(e, inv_pitch);
                const int col = e - r * pitch;
                const unsigned off = 4u * (unsigned)(clampi(by0 + r, 0, h - 1) * hs + clampi(bx0 + col, 0, w - 1));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (db_lptr_t)(slot + e0), 4, (col < bw && r < bh) ? off : 0x80000000u, 0, 0, 0);
            }
        }
        for (int e = tid; e < ncell * cn; e += DB_THREADS) cells[e] = 0ull;
        float gv[DB_CH];
#pragma unroll
        for (int cc = 0; cc < DB_CH; ++cc) gv[cc] = (valid && cc < cn) ? gpx[(int64_t)(c0 + cc) * s1.c] : 0.0f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        if (valid) {
            // taps outside, the pass's channels inside: a tap's corner index, fractions, products of fractions and quadrant
            // masks are formed once for the three channels.  Every sum keeps its order: a tap's filter / offset gradient
            // sums visit the channels in increasing order, a channel's quadrant sums the taps in increasing order.
            float qg[DB_CH][4], q[DB_CH][4];
#pragma unroll
            for (int cc = 0; cc < DB_CH; ++cc) {
                const float g = gv[cc];
                qg[cc][0] = g * (1.0f - alpha) * (1.0f - beta); qg[cc][1] = g * alpha * (1.0f - beta);
                qg[cc][2] = g * (1.0f - alpha) * beta;          qg[cc][3] = g * alpha * beta;
                q[cc][0] = q[cc][1] = q[cc][2] = q[cc][3] = 0.0f;
            }
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int dj = k / FS, di = k % FS;
                int o = lb[k];
                float phiY = phy[k], phiX = phx[k];
                asm volatile("" : "+v"(o), "+v"(phiY), "+v"(phiX));    // (corner weights and addresses re-derived per pass: registers)
                const float wTL = (1.0f - phiX) * (1.0f - phiY), wTR = phiX * (1.0f - phiY), wBL = (1.0f - phiX) * phiY, wBR = phiY * phiX;
                // quadrant: by integer index (VARIANT 0) or by displaced position (1, 2)
                unsigned quad;
                if constexpr (VARIANT == VFI_DEFOR_OFFSET) quad = (dj >= FS / 2 ? 2u : 0u) + (di >= FS / 2 ? 1u : 0u);
                else quad = ((qx >> k) & 1u) | (((qy >> k) & 1u) << 1);
                const float kqq = quad == 0 ? kq[0] : quad == 1 ? kq[1] : quad == 2 ? kq[2] : kq[3];
                const int cell = crow[dj] + co[di];
#pragma unroll
                for (int cc = 0; cc < DB_CH; ++cc) {
                    if (cc < cn) {                          // (block-uniform)
                        const float g = gv[cc];
                        const float* t = wins + cc * DB_WIN_FLOATS + o;
                        const float vTL = t[0], vTR = t[1], vBL = t[pitch], vBR = t[pitch + 1];
                        float v = wTL * vTL;
                        v = fmaf(wTR, vTR, v);
                        v = fmaf(wBL, vBL, v);
                        v = fmaf(wBR, vBR, v);
                        float dY = (-(1.0f - phiX)) * vTL;
                        dY = fmaf(1.0f - phiX, vBL, dY);
                        dY = fmaf(-phiX, vTR, dY);
                        dY = fmaf(phiX, vBR, dY);
                        float dX = (-(1.0f - phiY)) * vTL;
                        dX = fmaf(1.0f - phiY, vTR, dX);
                        dX = fmaf(-phiY, vBL, dX);
                        dX = fmaf(phiY, vBR, dX);
                        const float qgq = quad == 0 ? qg[cc][0] : quad == 1 ? qg[cc][1] : quad == 2 ? qg[cc][2] : qg[cc][3];
                        unsigned long long* cp = &cells[cc * ncell + cell];
                        if constexpr (VARIANT == VFI_DEFOR_NOFILTER) {
                            atomicAdd(cp, (unsigned long long)__float2ll_rn(qgq * gctx.scale));
#pragma unroll
                            for (int u = 0; u < 4; ++u) q[cc][u] = quad == (unsigned)u ? q[cc][u] + v : q[cc][u];
                            goy[k] += g * kqq * dY;
                            gox[k] += g * kqq * dX;
                        } else {
                            const float wg = wgt[k];
                            atomicAdd(cp, (unsigned long long)__float2ll_rn(qgq * wg * gctx.scale));
                            gfa[k] += qgq * v;
#pragma unroll
                            for (int u = 0; u < 4; ++u) q[cc][u] = quad == (unsigned)u ? fmaf(v, wg, q[cc][u]) : q[cc][u];
                            goy[k] += g * kqq * dY * wg;
                            gox[k] += g * kqq * dX * wg;
                        }
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < DB_CH; ++cc) {
                if (cc < cn) {
                    const float g = gv[cc];
                    {
                        const float gamma = 1.0f - beta;
                        float temp = gamma * (q[cc][1] - q[cc][0]);
                        temp = fmaf(1.0f - gamma, q[cc][3] - q[cc][2], temp);
                        gx = fmaf(g, temp, gx);
                    }
                    {
                        const float gamma = 1.0f - alpha;
                        float temp = gamma * (q[cc][2] - q[cc][0]);
                        temp = fmaf(1.0f - gamma, q[cc][3] - q[cc][1], temp);
                        gy = fmaf(g, temp, gy);
                    }
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < ncell * cn; e += DB_THREADS) {
            const unsigned long long v = cells[e];
            if (v != 0ull) {
                const int cc = e / ncell, r = e - cc * ncell;
                const int cy = r / gw, cx = r - cy * gw;
                atomicAdd(&gimg[(int64_t)(c0 + cc) * h * w + (int64_t)(gy0 + cy) * w + gx0 + cx], v);
            }
        }
        __syncthreads();                                    // (the next pass overwrites windows and cells)
    }
    if (valid) {
        float* gf = g2 + (int64_t)b * s2.b + (int64_t)y * s2.h + x;
        gf[0] = gx;
        gf[s2.c] = gy;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            gopx[(int64_t)k * ocs] = goy[k];
            gopx[(int64_t)(NT + k) * ocs] = gox[k];
            if constexpr (VARIANT != VFI_DEFOR_NOFILTER) gfpx[(int64_t)k * s3.c] = gfa[k];
        }
    }
}

}  // namespace vfi

using namespace vfi;

// internal (filterinterp.hip): launches the staged backward of `variant` for fs == 4; blocks it flags are left to the
// caller's per-tap launch.  Returns -1 when the shape does not fit its 32-bit plane offsets (nothing launched, no flags).
extern "C" int vfi_filterinterp_backward_defor_lds(int variant, const float* input1, const float* input2, const float* input3,
                                                    const float* input4, const float* gradoutput, unsigned long long* acc,
                                                    const int* hdr, int* flags, float* gradinput2, float* gradinput3,
                                                    float* gradinput4, int batch, int channel, int h, int w, vfi_strides s1,
                                                    vfi_strides s2, vfi_strides s3, vfi_strides s4, vfi_stream_t stream) {
    if ((int64_t)h * s1.h * 4 > INT_MAX) return -1;          // byte offsets inside a plane are 32-bit
    const dim3 grid = pixel_grid(w, h, batch), block(DB_TW, DB_TH, 1);
    static_assert(DB_TW == VFI_TX && DB_TH == VFI_TY, "the per-tap kernel's blocks are this kernel's");
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
    case VFI_DEFOR_OFFSET:
        hipLaunchKernelGGL(fi_backward_defor_lds<VFI_DEFOR_OFFSET>, grid, block, 0, st, input1, input2, input3, input4, gradoutput, acc,
                           hdr, flags, gradinput2, gradinput3, gradinput4, channel, h, w, s1, s2, s3, s4);
        break;
    case VFI_DEFOR_REGION:
        hipLaunchKernelGGL(fi_backward_defor_lds<VFI_DEFOR_REGION>, grid, block, 0, st, input1, input2, input3, input4, gradoutput, acc,
                           hdr, flags, gradinput2, gradinput3, gradinput4, channel, h, w, s1, s2, s3, s4);
        break;
    default:
        hipLaunchKernelGGL(fi_backward_defor_lds<VFI_DEFOR_NOFILTER>, grid, block, 0, st, input1, input2, input3, input3, gradoutput, acc,
                           hdr, flags, gradinput2, gradinput3, gradinput3, channel, h, w, s1, s2, s3, s3);
        break;
    }
    return launch_status();
}
