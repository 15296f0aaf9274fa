// cxx_bench.cpp -- the C ABI of libvfi_hip.so used from plain C++ / HIP, no torch and no Python:
// FilterInterpolation (C = 196 and 3) and FlowProjection at 1080p (padded 1152x1984), timed with HIP
// events on the stream the library launches on.
//   hipcc --offload-arch=gfx950 -O2 -I include tools/cxx_bench.cpp -L <pkg>/lib -lvfi_hip -Wl,-rpath,<pkg>/lib -o /tmp/cxx_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vfi_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// smooth synthetic flow: a few low-frequency waves, a few pixels of amplitude
__global__ void fill_flow(float* f, int h, int w, float amp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    f[(size_t)y * w + x] = amp * (sinf(x * 0.013f + y * 0.007f) + 0.5f * sinf(y * 0.021f - x * 0.004f));
    f[(size_t)h * w + (size_t)y * w + x] = amp * (cosf(x * 0.009f - y * 0.011f) + 0.5f * sinf(x * 0.017f + 1.0f));
}
__global__ void fill_hash(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)i * 2654435761u ^ seed;
        v ^= v >> 15; v *= 2246822519u; v ^= v >> 13;
        p[i] = scale * (float)(v >> 8) / 16777216.0f;
    }
}

template <typename F>
static float time_ms(hipStream_t st, int iters, F&& f) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i) f();
    CHECK(hipEventRecord(b, st));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    const int h = 1152, w = 1984, C = 196;
    const size_t px = (size_t)h * w;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    float *ctx, *out, *flow, *filt, *count, *proj;
    CHECK(hipMalloc(&ctx, C * px * 4));
    CHECK(hipMalloc(&out, C * px * 4));
    CHECK(hipMalloc(&flow, 2 * px * 4));
    CHECK(hipMalloc(&filt, 16 * px * 4));
    CHECK(hipMalloc(&count, px * 4));
    CHECK(hipMalloc(&proj, 2 * px * 4));
    hipLaunchKernelGGL(fill_hash, dim3(4096), dim3(256), 0, st, ctx, C * px, 1u, 1.0f);
    hipLaunchKernelGGL(fill_hash, dim3(4096), dim3(256), 0, st, filt, 16 * px, 2u, 0.125f);
    hipLaunchKernelGGL(fill_flow, dim3((w + 255) / 256, h), dim3(256), 0, st, flow, h, w, 6.0f);
    CHECK(hipStreamSynchronize(st));
    printf("%s\n", vfi_version());
    const vfi_strides s196{(int64_t)C * px, (int64_t)px, w}, s3{(int64_t)3 * px, (int64_t)px, w};
    const vfi_strides s2{(int64_t)2 * px, (int64_t)px, w}, s16{(int64_t)16 * px, (int64_t)px, w}, s1{(int64_t)px, (int64_t)px, w};
    int err = 0;
    float ms = time_ms(st, 20, [&] { err |= vfi_filterinterp_forward_ori(ctx, flow, filt, out, 1, C, h, w, 16, s196, s2, s16, st); });
    printf("FilterInterpolation C=196  %8.4f ms  %7.1f GB/s (1640 B/px)  err %d\n", ms, 1640.0 * px / ms / 1e6, err);
    ms = time_ms(st, 100, [&] { err |= vfi_filterinterp_forward_ori(ctx, flow, filt, out, 1, 3, h, w, 16, s3, s2, s16, st); });
    printf("FilterInterpolation C=3    %8.4f ms  %7.1f GB/s (96 B/px)    err %d\n", ms, 96.0 * px / ms / 1e6, err);
    ms = time_ms(st, 100, [&] { err |= vfi_flowprojection_forward(flow, count, proj, 1, h, w, 1, s2, s1, st); });
    printf("FlowProjection (fillhole)  %8.4f ms  %7.1f GB/s (20 B/px)    err %d\n", ms, 20.0 * px / ms / 1e6, err);
    return err;
}
