// Probe: throughput of ds_add_f32 (no return) on gfx950 for the access shapes of the projection
// kernel: 64 distinct consecutive cells, few active lanes, and colliding lanes.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/probes/lds_atomic_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 2000
#define THREADS 256

template <int MODE>
__global__ __launch_bounds__(THREADS) void probe(float* __restrict__ out, int stride) {
    __shared__ float acc[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += THREADS) acc[i] = 0.0f;
    __syncthreads();
    int a = (tid * stride) & 4095;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {                                // all 64 lanes, distinct consecutive cells
            atomicAdd(&acc[a], 1.0f);
        } else if (MODE == 1) {                         // 8 active lanes per wave
            if (lane < 8) atomicAdd(&acc[a], 1.0f);
        } else if (MODE == 2) {                         // pairs of lanes on one cell
            atomicAdd(&acc[a & ~1], 1.0f);
        } else if (MODE == 3) {                         // plain store for reference
            acc[a] = (float)it;
        } else if (MODE == 4) {                         // 3 planes like the projection (x, y, count)
            atomicAdd(&acc[a & 1023], 1.0f);
            atomicAdd(&acc[1024 + (a & 1023)], 2.0f);
            atomicAdd(&acc[2048 + (a & 1023)], 3.0f);
        }
        a = (a + 65) & 4095;
    }
    __syncthreads();
    out[blockIdx.x * THREADS + tid] = acc[tid];
}

int main() {
    float* dout;
    const int blocks = 2048;
    hipMalloc(&dout, blocks * THREADS * 4);
    const char* names[5] = {"64 lanes distinct", "8 lanes active", "lane pairs collide", "plain ds_write", "3 planes x 64 lanes"};
    for (int mode = 0; mode < 5; ++mode) {
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
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double winstr = (double)blocks * (THREADS / 64) * ITERS * (mode == 4 ? 3 : 1);
        // 256 CUs: cycles per wave-instruction per CU at 2.1 GHz
        printf("mode %d %-22s %8.3f ms  %7.2f G wave-instr/s  ~%5.1f cycles/instr/CU\n", mode, names[mode], ms,
               winstr / ms / 1e6, ms * 1e-3 * 2.1e9 * 256 / winstr);
    }
    return 0;
}
