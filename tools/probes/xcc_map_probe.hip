// Probe (round 3): what a three-launch chain like the projection's can rely on, for speed only.
//   1. workgroup -> XCD dealing across consecutive launches of different grid sizes: is (xcc - block) mod 8 the
//      same for every block of a launch, and does it stay the same from launch to launch?
//   2. what a launch of N workgroups of T threads costs when every workgroup reads one word and leaves
//      (proj_finish on a frame without holes);
//   3. a second kernel re-reading what a first kernel has just read: same XCD (L2 hit) against another XCD
//      (Infinity Cache hit).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/xcc_map_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void who(int* __restrict__ xcc_of_block) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = (int)(xcc & 15);
}

__global__ void one_word(const int* __restrict__ flags, int* __restrict__ sink) {
    if (flags[blockIdx.x] != 0) sink[blockIdx.x * blockDim.x + threadIdx.x] = 1;
}

// every workgroup reads a 16 KB piece (256 threads x 4 x 16 B); piece = f(block): shift rotates the XCD that reads it
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void reader(const v4f* __restrict__ src, float* __restrict__ sink, int pieces, int shift) {
    const int b = blockIdx.x;
    // piece p is read by block p in the first kernel; with shift s by block (p + s): another XCD when s % 8 != 0
    int p = b - shift;
    if (p < 0) p += pieces;
    const v4f* s = src + (size_t)p * 1024 + threadIdx.x;
    v4f a = s[0], c = s[256], d = s[512], e = s[768];
    const float t = a.x + c.y + d.z + e.w + a.w + c.x;
    if (t == 123456.789f) sink[b] = t;
}

static float time_launches(void (*launch)(void*), void* ctx, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch(ctx);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch(ctx);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

struct OneWord { const int* flags; int* sink; int blocks, threads; };
static void launch_one_word(void* c) {
    OneWord* o = (OneWord*)c;
    hipLaunchKernelGGL(one_word, dim3(o->blocks), dim3(o->threads), 0, 0, o->flags, o->sink);
}
struct Pair { const v4f* src; float* sink; int pieces, shift; };
static void launch_pair(void* c) {
    Pair* p = (Pair*)c;
    hipLaunchKernelGGL(reader, dim3(p->pieces), dim3(256), 0, 0, p->src, p->sink, p->pieces, 0);
    hipLaunchKernelGGL(reader, dim3(p->pieces), dim3(256), 0, 0, p->src, p->sink, p->pieces, p->shift);
}
static void launch_single(void* c) {
    Pair* p = (Pair*)c;
    hipLaunchKernelGGL(reader, dim3(p->pieces), dim3(256), 0, 0, p->src, p->sink, p->pieces, 0);
}

int main() {
    int* dmap;
    hipMalloc(&dmap, 1 << 20);
    // ---- 1. dealing
    const int grids[] = {558, 2232, 2232, 279, 1000, 2232, 558, 2232, 2232, 64, 2232, 2233, 2232};
    const int threads[] = {256, 128, 1024, 64, 512, 128, 256, 128, 256, 1024, 128, 128, 128};
    std::vector<int> h(4096);
    for (int rep = 0; rep < 2; ++rep)
        for (size_t k = 0; k < sizeof(grids) / sizeof(int); ++k) {
            hipLaunchKernelGGL(who, dim3(grids[k]), dim3(threads[k]), 0, 0, dmap);
            hipMemcpy(h.data(), dmap, grids[k] * 4, hipMemcpyDeviceToHost);
            int rot = ((h[0] - 0) % 8 + 8) % 8, bad = 0;
            for (int b = 0; b < grids[k]; ++b)
                if (((h[b] - b) % 8 + 8) % 8 != rot) ++bad;
            printf("grid %5d x %4d threads: block 0 on xcc %d, blocks off the round robin: %d\n", grids[k], threads[k], h[0], bad);
        }
    // the same without the synchronising copy in between: a chain of three launches, maps read afterwards
    {
        int *m0, *m1, *m2;
        hipMalloc(&m0, 16384); hipMalloc(&m1, 16384); hipMalloc(&m2, 16384);
        for (int rep = 0; rep < 4; ++rep) {
            hipLaunchKernelGGL(who, dim3(558), dim3(256), 0, 0, m0);
            hipLaunchKernelGGL(who, dim3(2232), dim3(128), 0, 0, m1);
            hipLaunchKernelGGL(who, dim3(2232), dim3(256), 0, 0, m2);
            hipDeviceSynchronize();
            int a, b, c;
            hipMemcpy(&a, m0, 4, hipMemcpyDeviceToHost); hipMemcpy(&b, m1, 4, hipMemcpyDeviceToHost); hipMemcpy(&c, m2, 4, hipMemcpyDeviceToHost);
            printf("chain 558 -> 2232 -> 2232: block 0 on xcc %d, %d, %d\n", a, b, c);
        }
    }
    // ---- 2. a launch whose workgroups read one word and leave
    int* flags; int* sink;
    hipMalloc(&flags, 1 << 16); hipMemset(flags, 0, 1 << 16);
    hipMalloc(&sink, 2232 * 1024 * 4);
    const int shapes[][2] = {{2232, 1024}, {2232, 512}, {2232, 256}, {2232, 128}, {2232, 64}, {558, 1024}, {558, 256}, {279, 1024}, {256, 1024}, {35, 64}};
    for (auto& s : shapes) {
        OneWord o{flags, sink, s[0], s[1]};
        printf("one word per workgroup, %5d x %4d threads: %6.2f us per launch (back to back)\n", s[0], s[1], time_launches(launch_one_word, &o, 300));
    }
    // ---- 3. re-read on the same / another XCD.  36.6 MB = 2232 pieces of 16 KB
    {
        const int pieces = 2232;
        v4f* src; float* fs;
        hipMalloc(&src, (size_t)pieces * 16384 * 12);
        hipMemset(src, 0, (size_t)pieces * 16384 * 12);
        hipMalloc(&fs, 1 << 16);
        Pair single{src, fs, pieces, 0};
        printf("one reader of %.1f MB alone (hot): %6.2f us\n", pieces * 16384 / 1e6, time_launches(launch_single, &single, 300));
        for (int shift : {0, 8, 16, 1, 3, 4, 279 * 4}) {
            Pair p{src, fs, pieces, shift};
            printf("reader then re-reader shifted by %4d blocks: %6.2f us per pair (hot)\n", shift, time_launches(launch_pair, &p, 300));
        }
        // cold: rotate over 12 buffers (440 MB)
        for (int shift : {0, 8, 1, 4}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 240;
            for (int w = 0; w < 2; ++w) {
                hipEventRecord(e0);
                for (int i = 0; i < iters; ++i) {
                    Pair p{src + (size_t)(i % 12) * pieces * 1024, fs, pieces, shift};
                    launch_pair(&p);
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("cold: reader then re-reader shifted by %4d blocks: %6.2f us per pair\n", shift, ms * 1e3f / iters);
        }
    }
    return 0;
}
