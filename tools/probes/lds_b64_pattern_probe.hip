// Probe: LDS cycles of ds_read_b64 / ds_read_b32 under the address patterns of the FilterInterpolation tap
// gather (pairs shared by neighbouring lanes, stretched columns, a row change inside the wave).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_b64_pattern_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 12288
#define ITERS 2000
#define THREADS 512

typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float* lds_ptr;
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)(lds_ptr)p; }

template <int WIDE>
__global__ __launch_bounds__(THREADS) void probe(const float* __restrict__ in, const int* __restrict__ idx,
                                                 float* __restrict__ out) {
    __shared__ float lds[N];
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += THREADS) lds[i] = in[i];
    __syncthreads();
    const unsigned a = lds_addr(lds + idx[tid]);
    float acc = 0.0f;
    for (int it = 0; it < ITERS; ++it) {
        if (WIDE) {
            f2 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile(
                "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\t"
                "ds_read_b64 %3, %8 offset:384\n\tds_read_b64 %4, %8 offset:392\n\tds_read_b64 %5, %8 offset:400\n\t"
                "ds_read_b64 %6, %8 offset:768\n\tds_read_b64 %7, %8 offset:776\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                : "v"(a) : "memory");
            acc += r0.x + r1.y + r2.x + r3.y + r4.x + r5.y + r6.x + r7.y;
        } else {
            float r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile(
                "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\t"
                "ds_read_b32 %3, %8 offset:12\n\tds_read_b32 %4, %8 offset:384\n\tds_read_b32 %5, %8 offset:388\n\t"
                "ds_read_b32 %6, %8 offset:392\n\tds_read_b32 %7, %8 offset:396\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                : "v"(a) : "memory");
            acc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
        }
    }
    out[blockIdx.x * THREADS + tid] = acc;
}

int main() {
    std::vector<float> h(N, 1.0f);
    float *din, *dout;
    int* didx;
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 2;
    hipMalloc(&din, N * 4);
    hipMalloc(&dout, blocks * THREADS * 4);
    hipMalloc(&didx, THREADS * 4);
    hipMemcpy(din, h.data(), N * 4, hipMemcpyHostToDevice);
    const char* names[] = {"b64 lane->pair lane (all distinct)", "b64 lane->pair lane/2 (shared by 2)",
                           "b64 FI even/odd origins, uniform shift", "b64 FI stretched (+1 col every 12 lanes)",
                           "b64 FI stretched + row change at lane 20/45, pitch 96", "b64 same, pitch 64",
                           "b32 lane->float lane", "b32 stretched", "b32 stretched + row change, pitch 96"};
    for (int pat = 0; pat < 9; ++pat) {
        std::vector<int> idx(THREADS);
        for (int t = 0; t < THREADS; ++t) {
            const int lane = t & 63, wave = t >> 6;
            const int row0 = wave * 3;
            int col = lane, row = row0, pitch = 96;
            switch (pat) {
            case 0: col = 2 * lane; break;
            case 1: col = lane & ~1; break;
            case 2: col = (lane + 5) & ~1; break;
            case 3: col = (lane + 5 + lane / 12) & ~1; break;
            case 4: col = (lane + 5 + lane / 12) & ~1; row = row0 + (lane >= 20) + (lane >= 45); break;
            case 5: col = (lane + 5 + lane / 12) & ~1; row = row0 + (lane >= 20) + (lane >= 45); pitch = 64; break;
            case 6: col = lane; break;
            case 7: col = lane + 5 + lane / 12; break;
            case 8: col = lane + 5 + lane / 12; row = row0 + (lane >= 20) + (lane >= 45); break;
            }
            idx[t] = row * pitch + col;
        }
        hipMemcpy(didx, idx.data(), THREADS * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (pat < 6) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(THREADS), 0, 0, din, didx, dout);
            else hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(THREADS), 0, 0, din, didx, dout);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        // wave-instructions per CU = 2 blocks * 8 waves * ITERS * 8
        const double instr = 2.0 * 8 * ITERS * 8;
        printf("%-58s %8.3f ms  %6.2f ns per wave-instr per CU\n", names[pat], ms, ms * 1e6 / instr);
    }
    return 0;
}
