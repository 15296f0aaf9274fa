// Probe: what HBM rate does the FilterInterpolation C=196 walk itself allow?  Workgroups own a 64 x TH pixel tile
// and march through all planes (read 1 float per pixel, write 1 float per pixel), no LDS, no barrier, no halo.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/tile_walk_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int TH, int UNROLL>
__global__ __launch_bounds__(512) void walk(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W,
                                            int tiles_x) {
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = tx * 64 + (threadIdx.x & 63);
    const int r = threadIdx.x >> 6;                       // 8 waves
    constexpr int PX = TH / 8;
    const size_t cs = (size_t)H * W;
    int off[PX];
    bool ok[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int y = ty * TH + r + p * 8;
        ok[p] = x < W && y < H;
        off[p] = ok[p] ? y * W + x : 0;
    }
    for (int c = 0; c < C; c += UNROLL) {
        float v[UNROLL][PX];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int p = 0; p < PX; ++p) v[u][p] = (c + u < C) ? in[(size_t)(c + u) * cs + off[p]] : 0.0f;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int p = 0; p < PX; ++p)
                if (ok[p] && c + u < C) out[(size_t)(c + u) * cs + off[p]] = v[u][p] + 1.0f;
    }
}

// the same bytes as one flat stream (float4 per thread, grid-stride)
__global__ __launch_bounds__(256) void flat(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i];
        v.x += 1.0f; v.y += 1.0f; v.z += 1.0f; v.w += 1.0f;
        out[i] = v;
    }
}

template <typename F>
static float timeit(F&& f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 10;
}

int main() {
    const int C = 196, H = 1152, W = 1984;
    const size_t n = (size_t)C * H * W;
    float *in, *out;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, n * 4);
    hipMemset(in, 0, n * 4);
    const double gb = 2.0 * n * 4 / 1e9;
    const int tiles_x = (W + 63) / 64;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(flat, dim3(256 * 16), dim3(256), 0, 0, (const float4*)in, (float4*)out, n / 4); });
    printf("flat float4 stream              %7.3f ms %7.1f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL((walk<16, 1>), dim3(tiles_x * ((H + 15) / 16)), dim3(512), 0, 0, in, out, C, H, W, tiles_x); });
    printf("tile 64x16 walk, 1 plane/iter   %7.3f ms %7.1f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL((walk<16, 4>), dim3(tiles_x * ((H + 15) / 16)), dim3(512), 0, 0, in, out, C, H, W, tiles_x); });
    printf("tile 64x16 walk, 4 planes/iter  %7.3f ms %7.1f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL((walk<16, 8>), dim3(tiles_x * ((H + 15) / 16)), dim3(512), 0, 0, in, out, C, H, W, tiles_x); });
    printf("tile 64x16 walk, 8 planes/iter  %7.3f ms %7.1f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL((walk<32, 4>), dim3(tiles_x * ((H + 31) / 32)), dim3(512), 0, 0, in, out, C, H, W, tiles_x); });
    printf("tile 64x32 walk, 4 planes/iter  %7.3f ms %7.1f GB/s\n", ms, gb / ms * 1e3);
    ms = timeit([&] { hipLaunchKernelGGL((walk<8, 8>), dim3(tiles_x * ((H + 7) / 8)), dim3(512), 0, 0, in, out, C, H, W, tiles_x); });
    printf("tile 64x8 walk, 8 planes/iter   %7.3f ms %7.1f GB/s\n", ms, gb / ms * 1e3);
    return 0;
}
