// Probe: are 4-byte-aligned ds_read_b64 / ds_read_b128 usable on gfx950 under ROCm (unaligned
// DS access mode), and what do they cost next to ds_read2_b32 for a "4 consecutive floats at an
// arbitrary dword address" gather (one window row of FilterInterpolation)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_unaligned_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define N 8192
#define ITERS 4000
#define THREADS 512

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float* lds_ptr;

__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)(lds_ptr)p; }

template <int MODE>
__global__ __launch_bounds__(THREADS) void probe(const float* __restrict__ in, const int* __restrict__ idx,
                                                 float* __restrict__ out, int step) {
    __shared__ float lds[N];
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += THREADS) lds[i] = in[i];
    __syncthreads();
    int a = idx[tid];
    float acc = 0.0f;
    for (int it = 0; it < ITERS; ++it) {
        float v0, v1, v2, v3;
        if (MODE == 0) {                    // compiler: 2 x ds_read2_b32
            const float* p = lds + a;
            v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3];
        } else if (MODE == 1) {             // one ds_read_b128 at a dword-aligned address
            f4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(lds_addr(lds + a)) : "memory");
            v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w;
        } else {                            // two ds_read_b64 at dword-aligned addresses
            f2 lo, hi;
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(lo), "=&v"(hi) : "v"(lds_addr(lds + a)) : "memory");
            v0 = lo.x; v1 = lo.y; v2 = hi.x; v3 = hi.y;
        }
        acc += v0 + 2.0f * v1 + 3.0f * v2 + 4.0f * v3;
        a = (a + step) & (N - 1);
        if (a > N - 4) a -= 4;
    }
    out[blockIdx.x * THREADS + tid] = acc;
}

int main() {
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)(i % 251);
    float *din, *dout;
    int* didx;
    const int blocks = 512;
    hipMalloc(&din, N * 4);
    hipMalloc(&dout, blocks * THREADS * 4);
    hipMalloc(&didx, THREADS * 4);
    hipMemcpy(din, h.data(), N * 4, hipMemcpyHostToDevice);
    // pattern: lane i of a wave reads 4 floats starting at column (lane + jitter) of a row that
    // changes every 8 lanes; row pitch 96 floats (a multiple of 32 banks); alignment cases:
    // shift 0 -> 16-B aligned only for lanes % 4 == 0, so every alignment class occurs
    for (int shift = 0; shift < 2; ++shift) {
        std::vector<int> idx(THREADS);
        for (int t = 0; t < THREADS; ++t) {
            const int lane = t & 63, wave = t >> 6;
            idx[t] = ((wave * 5 + (lane / 8)) % 20) * 96 + lane + shift * ((lane / 16) & 1);
        }
        hipMemcpy(didx, idx.data(), THREADS * 4, hipMemcpyHostToDevice);
        // expected result computed on the host
        std::vector<float> ref(THREADS), got(THREADS);
        for (int t = 0; t < THREADS; ++t) {
            int a = idx[t];
            float acc = 0;
            for (int it = 0; it < ITERS; ++it) {
                acc += h[a] + 2.0f * h[a + 1] + 3.0f * h[a + 2] + 4.0f * h[a + 3];
                a = (a + 97) & (N - 1);
                if (a > N - 4) a -= 4;
            }
            ref[t] = acc;
        }
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(THREADS), 0, 0, din, didx, dout, 97);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(THREADS), 0, 0, din, didx, dout, 97);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(THREADS), 0, 0, din, didx, dout, 97);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(got.data(), dout, THREADS * 4, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int t = 0; t < THREADS; ++t) bad += (got[t] != ref[t]);
            const double reads = (double)blocks * THREADS * ITERS;
            printf("shift %d mode %d (%s): %8.3f ms  %6.2f Grows/s  mismatches %d/%d\n", shift, mode,
                   mode == 0 ? "2x ds_read2_b32" : mode == 1 ? "ds_read_b128    " : "2x ds_read_b64  ", ms,
                   reads / ms / 1e6, bad, THREADS);
        }
    }
    return 0;
}
