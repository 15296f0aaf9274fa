// Probe: do buffer_load_dwordx4 / buffer_store_dwordx4 / global dwordx4 work at addresses that are only 4-byte aligned?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const int* in, int* out_buf, int* out_glb, int* st_buf, int n, int shift) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)st_buf, 0, n * 4, 0x00020000);
    const int t = threadIdx.x;
    v4i a = __builtin_amdgcn_raw_buffer_load_b128(r, (4 * t + shift) * 4, 0, 0);
    out_buf[4 * t + 0] = a.x; out_buf[4 * t + 1] = a.y; out_buf[4 * t + 2] = a.z; out_buf[4 * t + 3] = a.w;
    const v4i g = *reinterpret_cast<const v4i*>(in + 4 * t + shift);
    out_glb[4 * t + 0] = g.x; out_glb[4 * t + 1] = g.y; out_glb[4 * t + 2] = g.z; out_glb[4 * t + 3] = g.w;
    v4i s = {1000 + 4 * t, 1001 + 4 * t, 1002 + 4 * t, 1003 + 4 * t};
    __builtin_amdgcn_raw_buffer_store_b128(s, w, (4 * t + shift) * 4, 0, 0);
}
int main() {
    const int n = 1024;
    int *in, *ob, *og, *sb;
    hipMalloc(&in, n * 4); hipMalloc(&ob, n * 4); hipMalloc(&og, n * 4); hipMalloc(&sb, n * 4);
    int h[n];
    for (int i = 0; i < n; ++i) h[i] = i;
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipMemset(sb, 0xff, n * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, in, ob, og, sb, n, shift);
        int a[256], b[256], c[300];
        hipMemcpy(a, ob, 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(b, og, 256 * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c, sb, 300 * 4, hipMemcpyDeviceToHost);
        int badb = 0, badg = 0, bads = 0;
        for (int i = 0; i < 256; ++i) { badb += a[i] != i + shift; badg += b[i] != i + shift; bads += c[i + shift] != 1000 + i; }
        printf("shift %d: buffer load bad %d (first %d %d %d %d), global load bad %d, buffer store bad %d (first %d %d %d %d %d)\n", shift, badb, a[0], a[1],
               a[2], a[3], badg, bads, c[0], c[1], c[2], c[3], c[4]);
    }
    return 0;
}
