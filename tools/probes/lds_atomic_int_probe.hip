// Probe: ds_add_u32 / ds_add_u64 (no return) next to ds_add_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 2000
#define THREADS 256

template <int MODE>
__global__ __launch_bounds__(THREADS) void probe(unsigned long long* __restrict__ out, int stride) {
    __shared__ unsigned long long acc64[2048];
    unsigned* acc32 = reinterpret_cast<unsigned*>(acc64);
    float* accf = reinterpret_cast<float*>(acc64);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2048; i += THREADS) acc64[i] = 0;
    __syncthreads();
    int a = (tid * stride) & 2047;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) atomicAdd(&acc32[a], 3u);                          // u32, 64 distinct lanes
        else if (MODE == 1) atomicAdd(&acc64[a], 3ull);                   // u64, 64 distinct lanes
        else if (MODE == 2) atomicAdd(&accf[a], 1.0f);                    // f32 for reference
        else if (MODE == 3) atomicAdd(&acc64[a & ~3], 3ull);              // u64, groups of 4 lanes collide
        else if (MODE == 4) { if (lane < 8) atomicAdd(&acc64[a], 3ull); } // u64, 8 active lanes
        else if (MODE == 5) atomicAdd(&acc32[a & ~3], 3u);                // u32, groups of 4 collide
        a = (a + 65) & 2047;
    }
    __syncthreads();
    out[blockIdx.x * THREADS + tid] = acc64[tid];
}

int main() {
    unsigned long long* dout;
    const int blocks = 2048;
    hipMalloc(&dout, blocks * THREADS * 8);
    const char* names[6] = {"u32 distinct", "u64 distinct", "f32 distinct", "u64 4-way collide", "u64 8 lanes", "u32 4-way collide"};
    for (int mode = 0; mode < 6; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(THREADS), 0, 0, dout, 1);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(THREADS), 0, 0, dout, 1);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(THREADS), 0, 0, dout, 1);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(THREADS), 0, 0, dout, 1);
            if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(THREADS), 0, 0, dout, 1);
            if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(THREADS), 0, 0, dout, 1);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double winstr = (double)blocks * (THREADS / 64) * ITERS;
        printf("mode %d %-20s %8.3f ms  ~%6.1f cycles/instr/CU\n", mode, names[mode], ms, ms * 1e-3 * 2.1e9 * 256 / winstr);
    }
    return 0;
}
