// Probe 2: which walk through a [C][H][W] tensor does HBM like?  Tile shape, phase of the plane walk, reads vs writes.
#include <hip/hip_runtime.h>
#include <cstdio>

// tile TW x TH, 512 threads, PX = TW*TH/512 pixels per thread; MODE 0 copy, 1 read only, 2 write only
template <int TW, int TH, int UNROLL, int MODE, int NT = 0>
__global__ __launch_bounds__(512) void walk(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W,
                                            int tiles_x, int phase_mul, int chunk) {
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    constexpr int PX = TW * TH / 512;
    constexpr int ROWS_PER_PASS = 512 / TW;
    const int x = tx * TW + (threadIdx.x % TW);
    const int r = threadIdx.x / TW;
    const size_t cs = (size_t)H * W;
    int off[PX];
    bool ok[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int y = ty * TH + r + p * ROWS_PER_PASS;
        ok[p] = x < W && y < H;
        off[p] = ok[p] ? y * W + x : 0;
    }
    const int c_lo = blockIdx.y * chunk, c_n = min(chunk, C - c_lo);
    const int c0 = (int)(((long long)tile * phase_mul) % c_n);
    float acc = 0.0f;
    for (int i = 0; i < c_n; i += UNROLL) {
        float v[UNROLL][PX];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int c = c_lo + (c0 + i + u) % c_n;
#pragma unroll
            for (int p = 0; p < PX; ++p) v[u][p] = (MODE != 2 && i + u < c_n) ? ((NT & 1) ? __builtin_nontemporal_load(in + (size_t)c * cs + off[p]) : in[(size_t)c * cs + off[p]]) : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int c = c_lo + (c0 + i + u) % c_n;
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                if (MODE == 1) acc += v[u][p];
                else if (ok[p] && i + u < c_n) { if (NT & 2) __builtin_nontemporal_store(v[u][p] + 1.0f, out + (size_t)c * cs + off[p]); else out[(size_t)c * cs + off[p]] = v[u][p] + 1.0f; }
            }
        }
    }
    if (MODE == 1 && acc == 123.456f) out[0] = acc;
}

// float4 per lane: tile (64*4) x TH... each thread 4 consecutive pixels of PX rows
// LDSPAD > 0: a dummy LDS allocation that caps the workgroups per CU (occupancy experiment)
template <int TH, int UNROLL, int NT, int LDSPAD = 0>
__global__ __launch_bounds__(512) void walk4(const float4* __restrict__ in, float4* __restrict__ out, int C, int H, int W4,
                                             int tiles_x) {
    __shared__ float pad[LDSPAD > 0 ? LDSPAD : 1];
    if (LDSPAD > 0 && C < 0) pad[threadIdx.x] = 1.0f;
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    constexpr int PX = TH / 8;
    const int x = tx * 64 + (threadIdx.x & 63);
    const int r = threadIdx.x >> 6;
    const size_t cs = (size_t)H * W4;
    int off[PX];
    bool ok[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int y = ty * TH + r + p * 8;
        ok[p] = x < W4 && y < H;
        off[p] = ok[p] ? y * W4 + x : 0;
    }
    for (int c = 0; c < C; c += UNROLL) {
        float4 v[UNROLL][PX];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int p = 0; p < PX; ++p) v[u][p] = (c + u < C) ? in[(size_t)(c + u) * cs + off[p]] : float4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int p = 0; p < PX; ++p)
                if (ok[p] && c + u < C) { float4 t = v[u][p]; t.x += 1.0f; out[(size_t)(c + u) * cs + off[p]] = t; }
    }
}

// float4 lanes on a 64-pixel-wide tile: 16 lanes per row (256-byte pieces), 4 rows per wave instruction
template <int TH, int UNROLL>
__global__ __launch_bounds__(512) void walk4n(const float4* __restrict__ in, float4* __restrict__ out, int C, int H, int W4,
                                              int tiles_x) {
    const int tile = blockIdx.x;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    constexpr int PX = TH / 32;                             // 512 threads = 16 lanes x 32 rows per pass
    const int x = tx * 16 + (threadIdx.x & 15);
    const int r = threadIdx.x >> 4;
    const size_t cs = (size_t)H * W4;
    int off[PX];
    bool ok[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int y = ty * TH + r + p * 32;
        ok[p] = x < W4 && y < H;
        off[p] = ok[p] ? y * W4 + x : 0;
    }
    for (int c = 0; c < C; c += UNROLL) {
        float4 v[UNROLL][PX];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int p = 0; p < PX; ++p) v[u][p] = (c + u < C) ? in[(size_t)(c + u) * cs + off[p]] : float4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int p = 0; p < PX; ++p)
                if (ok[p] && c + u < C) { float4 t = v[u][p]; t.x += 1.0f; out[(size_t)(c + u) * cs + off[p]] = t; }
    }
}

template <typename F>
static float timeit(F&& f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < 8; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 8;
}

#define RUN(NAME, TW, TH, UN, MODE, PHASE, CHUNK) RUNX(NAME, TW, TH, UN, MODE, PHASE, CHUNK, 0)
#define RUNX(NAME, TW, TH, UN, MODE, PHASE, CHUNK, NT)                                                                  \
    do {                                                                                                             \
        const int txs = (W + TW - 1) / TW, tys = (H + TH - 1) / TH;                                                  \
        const int groups = (C + CHUNK - 1) / CHUNK;                                                                  \
        float ms = timeit([&] { hipLaunchKernelGGL((walk<TW, TH, UN, MODE, NT>), dim3(txs * tys, groups), dim3(512), 0, 0, \
                                                   in, out, C, H, W, txs, PHASE, CHUNK); });                         \
        const double bytes = (MODE == 0 ? 2.0 : 1.0) * n * 4 / 1e9;                                                  \
        printf("%-62s %7.3f ms %7.1f GB/s\n", NAME, ms, bytes / ms * 1e3);                                           \
    } while (0)

int main() {
    const int C = 196, H = 1152, W = 1984;
    const size_t n = (size_t)C * H * W;
    float *in, *out;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, n * 4);
    hipMemset(in, 0, n * 4);
    RUN("copy  64x16 u4", 64, 16, 4, 0, 0, 196);
    RUN("copy  64x16 u2", 64, 16, 2, 0, 0, 196);
    RUN("copy  64x16 u1", 64, 16, 1, 0, 0, 196);
    RUNX("copy  64x16 u4 nt loads", 64, 16, 4, 0, 0, 196, 1);
    RUNX("copy  64x16 u4 nt stores", 64, 16, 4, 0, 0, 196, 2);
    RUNX("copy  64x16 u4 nt both", 64, 16, 4, 0, 0, 196, 3);
    RUNX("copy  64x16 u2 nt both", 64, 16, 2, 0, 0, 196, 3);
    RUNX("copy  64x16 u1 nt both", 64, 16, 1, 0, 0, 196, 3);
    RUNX("copy  64x16 u2 nt stores", 64, 16, 2, 0, 0, 196, 2);
    {
        const int W4 = W / 4, txs = (W4 + 63) / 64;
        float ms = timeit([&] { hipLaunchKernelGGL((walk4<8, 1, 0>), dim3(txs * ((H + 7) / 8)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txs); });
        printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 256x8 u1", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL((walk4<8, 2, 0>), dim3(txs * ((H + 7) / 8)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txs); });
        printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 256x8 u2", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL((walk4<16, 1, 0>), dim3(txs * ((H + 15) / 16)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txs); });
        printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 256x16 u1", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL((walk4<8, 1, 0, 16000>), dim3(txs * ((H + 7) / 8)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txs); });
        printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 256x8 u1, 2 workgroups per CU (16 waves)", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL((walk4<8, 2, 0, 16000>), dim3(txs * ((H + 7) / 8)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txs); });
        printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 256x8 u2, 2 workgroups per CU", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
        ms = timeit([&] { hipLaunchKernelGGL((walk4<8, 4, 0, 16000>), dim3(txs * ((H + 7) / 8)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txs); });
        printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 256x8 u4, 2 workgroups per CU", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
        {
            const int txn = (W4 + 15) / 16;
            ms = timeit([&] { hipLaunchKernelGGL((walk4n<32, 1>), dim3(txn * ((H + 31) / 32)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txn); });
            printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 64x32 tile (256-byte row pieces) u1", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
            ms = timeit([&] { hipLaunchKernelGGL((walk4n<64, 1>), dim3(txn * ((H + 63) / 64)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txn); });
            printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 64x64 tile (256-byte row pieces) u1", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
        }
        ms = timeit([&] { hipLaunchKernelGGL((walk4<8, 1, 0, 10000>), dim3(txs * ((H + 7) / 8)), dim3(512), 0, 0, (const float4*)in, (float4*)out, C, H, W4, txs); });
        printf("%-62s %7.3f ms %7.1f GB/s\n", "copy float4 256x8 u1, 4 workgroups per CU (32 waves, LDS 40K)", ms, 2.0 * n * 4 / 1e9 / ms * 1e3);
    }
    return 0;
}
