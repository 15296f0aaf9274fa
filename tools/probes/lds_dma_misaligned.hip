// Probe: does an LDS-DMA dword load (buffer_load_dword ... lds) accept a global address that is 2 mod 4 (gfx950)?
// The fp16-storage FilterInterpolation kernel could then stage a second copy of its window shifted by one half, and every
// pixel would read its four taps of a row as two whole dwords (no v_alignbit).   hipcc --offload-arch=gfx950 -O3 lds_dma_misaligned.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void k(const unsigned short* in, unsigned* out, int shift_bytes) {
    __shared__ unsigned lds[256];
    lds[threadIdx.x] = 0xdeadbeefu;
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 4096, 0x00020000);
    const unsigned off = 4u * threadIdx.x + (unsigned)shift_bytes;
    if (threadIdx.x < 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 4, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
    // the same through a VGPR load, for comparison
    if (threadIdx.x < 64) out[256 + threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
}

// 16 bytes per lane
__global__ void k16(const unsigned short* in, unsigned* out, int shift_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned lds[256];
    lds[threadIdx.x] = 0xdeadbeefu;
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 4096, 0x00020000);
    const unsigned off = 16u * threadIdx.x + (unsigned)shift_bytes;
    if (threadIdx.x < 64) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}

int main() {
    std::vector<unsigned short> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = (unsigned short)(i + 1);
    unsigned short* din; unsigned* dout;
    hipMalloc(&din, 4096); hipMalloc(&dout, 512 * 4);
    hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    for (int shift : {0, 2, 1}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, din, dout, shift);
        std::vector<unsigned> o(512);
        if (hipMemcpy(o.data(), dout, 512 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("shift %d: launch failed\n", shift); return 1; }
        int bad_lds = 0, bad_vgpr = 0;
        for (int t = 0; t < 64; ++t) {
            unsigned want = 0;
            const unsigned char* p = (const unsigned char*)h.data() + 4 * t + shift;
            want = p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24);
            bad_lds += o[t] != want; bad_vgpr += o[256 + t] != want;
        }
        printf("global address = %d mod 4: LDS-DMA dword %s (%d of 64 wrong; lane 1 got %08x), VGPR dword load %s (%d wrong)\n", shift,
               bad_lds ? "WRONG" : "correct", bad_lds, o[1], bad_vgpr ? "WRONG" : "correct", bad_vgpr);
    }
    for (int shift : {0, 2, 4, 6}) {
        hipLaunchKernelGGL(k16, dim3(1), dim3(256), 0, 0, din, dout, shift);
        std::vector<unsigned> o(256);
        if (hipMemcpy(o.data(), dout, 256 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("x4 shift %d: launch failed\n", shift); return 1; }
        int bad = 0;
        for (int t = 0; t < 256; ++t) {
            const unsigned char* p = (const unsigned char*)h.data() + 4 * t + shift;
            const unsigned want = p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24);
            bad += o[t] != want;
        }
        printf("global address = %d mod 16: LDS-DMA of 16 bytes per lane %s (%d of 256 dwords wrong; dword 1 got %08x)\n", shift, bad ? "WRONG" : "correct", bad, o[1]);
    }
    return 0;
}
