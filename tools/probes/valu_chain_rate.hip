// Probe: cycles per vector instruction and SIMD for dependent chains of v_fma_f32 / v_pk_fma_f32 / v_fma_mix_f32, as a function of
// the independent chains per wave (ILP) and the waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_chain_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP, int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    v2f acc[CH];
    const v2f m = { seed, seed * 0.5f };
    unsigned h = __float_as_uint(seed) | 0x3c003c00u;
#pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = v2f{ (float)i + threadIdx.x, 1.0f };
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(m.x), "v"(m.y));
                else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(m), "v"(m));
                else asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i].x) : "v"(h), "v"(m.x));
            }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CH>
static void run(const char* name, float* out, int waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const int blocks = 256 * waves_per_simd;            // 256 CUs x 4 SIMDs x waves_per_simd waves, 4 waves per block
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    const double insts_per_simd = (double)iters * 8 * CH * waves_per_simd;
    printf("%-14s chains %d  waves/SIMD %d: %7.3f ms  -> %5.2f cycles per instruction and SIMD at 2.4 GHz\n", name, CH, waves_per_simd, best,
           best * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0, 1>("v_fma_f32", out, w); run<0, 2>("v_fma_f32", out, w); run<0, 4>("v_fma_f32", out, w);
        run<1, 1>("v_pk_fma_f32", out, w); run<1, 2>("v_pk_fma_f32", out, w); run<1, 4>("v_pk_fma_f32", out, w);
        run<2, 1>("v_fma_mix_f32", out, w); run<2, 2>("v_fma_mix_f32", out, w); run<2, 4>("v_fma_mix_f32", out, w);
    }
    return 0;
}
