// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ on gfx950 for the load flavours this library uses.
// Every kernel reads the SAME 1.79 GB tensor (196 planes of 1152 x 1984 floats) exactly once, 64 x 16 tiles,
// rows 256-byte aligned, no halo, marching through the planes the way fi_forward_ori_lds does:
//   calib_dma4     buffer_load_dword ... lds   (4 bytes per lane, the window staging instruction of the FI kernels)
//   calib_dma16    buffer_load_dwordx4 ... lds (16 bytes per lane)
//   calib_gld4     global_load_dword to registers
//   calib_gld16    global_load_dwordx4 to registers
// Run under  rocprofv3 --pmc FETCH_SIZE  and  --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum  (separate passes);
// the true byte count per launch is printed.  tools/calibrate_fetch.sh does both and prints the factors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define H 1152
#define W 1984
#define C 196
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BYTES>
__global__ __launch_bounds__(512) void calib_dma(const float* __restrict__ img, float* __restrict__ sink) {
    __shared__ float tile[2][1024];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % (W / 64), ty = blockIdx.x / (W / 64);
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
        const auto plane = __builtin_amdgcn_make_buffer_rsrc((void*)(img + (int64_t)c * H * W), 0, H * W * 4, 0x00020000);
        float* l = tile[c & 1];
        if (BYTES == 4) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = tid + k * 512, r = e >> 6, col = e & 63;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (lptr_t)(l + e), 4, 4u * ((ty * 16 + r) * W + tx * 64 + col), 0, 0, 0);
            }
        } else if (tid < 256) {
            const int e = tid * 4, r = e >> 6, col = e & 63;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(plane, (lptr_t)(l + e), 16, 4u * ((ty * 16 + r) * W + tx * 64 + col), 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += l[tid] + l[tid + 512];
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int BYTES>
__global__ __launch_bounds__(512) void calib_gld(const float* __restrict__ img, float* __restrict__ sink) {
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % (W / 64), ty = blockIdx.x / (W / 64);
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float* plane = img + (int64_t)c * H * W;
        if (BYTES == 4) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = tid + k * 512, r = e >> 6, col = e & 63;
                acc += plane[(ty * 16 + r) * W + tx * 64 + col];
            }
        } else if (tid < 256) {
            const int e = tid * 4, r = e >> 6, col = e & 63;
            const float4 v = *reinterpret_cast<const float4*>(plane + (ty * 16 + r) * W + tx * 64 + col);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    float *img, *sink;
    const size_t n = (size_t)C * H * W;
    if (hipMalloc(&img, n * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(img, 0, n * 4);
    const dim3 grid((W / 64) * (H / 16)), block(512);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_dma<4>, grid, block, 0, 0, img, sink);
        hipLaunchKernelGGL(calib_dma<16>, grid, block, 0, 0, img, sink);
        hipLaunchKernelGGL(calib_gld<4>, grid, block, 0, 0, img, sink);
        hipLaunchKernelGGL(calib_gld<16>, grid, block, 0, 0, img, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("true bytes read per launch: %zu\n", n * 4);
    return 0;
}
