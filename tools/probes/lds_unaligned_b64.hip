// Probe: ds_read_b64 at a 2-byte-aligned LDS address (gfx950, unaligned access mode): correctness for every byte offset
// parity, and the cost against the aligned read and against the ds_read2_b32 + ds_read_b32 + 2 x v_alignbit sequence of the
// fp16-storage FilterInterpolation kernel.   hipcc --offload-arch=gfx950 -O3 lds_unaligned_b64.hip -o /tmp/ua && /tmp/ua
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__global__ void check(const unsigned short* in, unsigned long long* out, int shift) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = in[i];
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + 2u * (unsigned)(threadIdx.x * 3 + shift);
    v2u d;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a));
    out[threadIdx.x] = ((unsigned long long)d.y << 32) | d.x;
}

template <int MODE>
__global__ void cost(unsigned* sink, int iters, int stride_halves, int shift) {
    __shared__ unsigned short lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + 2u * (unsigned)((threadIdx.x & 63) * stride_halves + shift * (threadIdx.x & 1));
    unsigned acc = 0;
    const unsigned sh = (a & 2) * 8;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            v2u d0, d1, d2, d3;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a));
            acc += d0.x ^ d0.y ^ d1.x ^ d1.y ^ d2.x ^ d2.y ^ d3.x ^ d3.y;
        } else {
            const unsigned al = a & ~3u;
            v2u d0, d1, d2, d3; unsigned e0, e1, e2, e3;
            asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:1\n\tds_read_b32 %4, %8 offset:8\n\t"
                         "ds_read2_b32 %1, %8 offset0:128 offset1:129\n\tds_read_b32 %5, %8 offset:520\n\t"
                         "ds_read2_b32 %2, %9 offset0:0 offset1:1\n\tds_read_b32 %6, %9 offset:8\n\t"
                         "ds_read2_b32 %3, %9 offset0:128 offset1:129\n\tds_read_b32 %7, %9 offset:520\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3), "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3) : "v"(al), "v"(al + 1024));
            acc += __builtin_amdgcn_alignbit(d0.y, d0.x, sh) ^ __builtin_amdgcn_alignbit(e0, d0.y, sh) ^ __builtin_amdgcn_alignbit(d1.y, d1.x, sh) ^
                   __builtin_amdgcn_alignbit(e1, d1.y, sh) ^ __builtin_amdgcn_alignbit(d2.y, d2.x, sh) ^ __builtin_amdgcn_alignbit(e2, d2.y, sh) ^
                   __builtin_amdgcn_alignbit(d3.y, d3.x, sh) ^ __builtin_amdgcn_alignbit(e3, d3.y, sh);
        }
        a ^= (acc & 0);
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    std::vector<unsigned short> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (unsigned short)(i * 7 + 1);
    unsigned short* din; unsigned long long* dout; unsigned* sink;
    hipMalloc(&din, 8192); hipMalloc(&dout, 256 * 8); hipMalloc(&sink, 4 * 256 * 512 * 8);
    hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
    int bad = 0;
    for (int shift = 0; shift < 8; ++shift) {
        hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, 0, din, dout, shift);
        std::vector<unsigned long long> o(256);
        hipMemcpy(o.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
        for (int t = 0; t < 256; ++t) {
            unsigned long long want = 0;
            for (int k = 3; k >= 0; --k) want = (want << 16) | h[t * 3 + shift + k];
            bad += o[t] != want;
        }
    }
    printf("ds_read_b64 at 2-byte-aligned addresses: %s (%d mismatches)\n", bad ? "WRONG" : "correct", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int stride : {4, 1, 2}) for (int shift : {0, 1}) for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(cost<0>, dim3(2048), dim3(512), 0, 0, sink, 2000, stride, shift);
            else hipLaunchKernelGGL(cost<1>, dim3(2048), dim3(512), 0, 0, sink, 2000, stride, shift);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
        }
        printf("stride %d halves, odd lanes shifted by %d half: %-34s %8.3f ms\n", stride, shift,
               mode == 0 ? "4 x ds_read_b64" : "4 x (read2_b32 + read_b32) + 8 alignbit", best);
    }
    return bad != 0;
}
