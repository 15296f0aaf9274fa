/*
 * vfi_oracle.h -- CPU restatement of the reference's frame-synthesis hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  The reference ships no
 * golden vectors, no tests and no CPU path for these ops (SURVEY.md section 4 and
 * 8c), and its CUDA sources cannot be compiled or run in this image (no nvcc,
 * no GPU, removed ATen APIs).  This restatement follows the .cu sources line
 * by line (citations on every function) and is pinned by (1) analytic
 * known-answer cases and (2) an independent vectorised numpy formulation
 * (oracle/np_oracle.py) -- see tests/test_oracle.py.
 *
 * All tensors are dense NCHW float32.  Every function returns 0 on success and
 * the reference binding's own error value (1) on a shape problem.
 *
 * `fmad`: 0 = C source semantics, no contraction (the library is compiled with
 * -ffp-contract=off); 1 = a*b+c accumulations fused the way nvcc's default
 * -fmad=true (and the HIP kernels in this repo) fuse them.  The two modes differ
 * by a few ulp; tests state which one they compare against.
 */
#ifndef VFI_ORACLE_H
#define VFI_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* threads for the OpenMP loops of A1 (when its nthreads argument is < 1) and A8 forward */
void vfi_oracle_set_num_threads(int n);
int vfi_oracle_get_num_threads(void);

/* A1  filterinterpolation_cuda_kernel.cu:2692-2823 (+ cc:537-606) */
int vfi_oracle_filterinterp_ori_fwd(const float* img, const float* flow, const float* filt,
                                    float* out, int B, int C, int H, int W, int filt_ch,
                                    int fmad, int nthreads);
/* A2  filterinterpolation_cuda_kernel.cu:2827-3125 */
int vfi_oracle_filterinterp_ori_bwd(const float* img, const float* flow, const float* filt,
                                    const float* gout, float* gimg, float* gflow, float* gfilt,
                                    int B, int C, int H, int W, int filt_ch, int fmad);

/* A1b/A1c/A1d  filterinterpolation_cuda_kernel.cu:29-426, 1353-1496, 2070-2191
 * variant 0: 4-input forward (quadrants by integer index, fs in {4,6} only)
 * variant 1: deforconv (quadrants by displaced position)
 * variant 2: nofilterwithdeforconv (variant 1 with unit weights; filt ignored)
 * off = [B, 2*fs*fs, H, W]: first fs*fs planes Y offsets, next fs*fs X offsets. */
int vfi_oracle_filterinterp_defor_fwd(int variant, const float* img, const float* flow,
                                      const float* filt, const float* off, float* out,
                                      int B, int C, int H, int W, int fs, int fmad);

/* backward of the same three variants (:430-1215, :1500-1935, :2195-2567); grads arrive zeroed;
 * variant 2: filt and gfilt are NULL. */
int vfi_oracle_filterinterp_defor_bwd(int variant, const float* img, const float* flow,
                                      const float* filt, const float* off, const float* gout,
                                      float* gimg, float* gflow, float* gfilt, float* goff,
                                      int B, int C, int H, int W, int fs, int fmad);

/* A3  flowprojection_cuda_kernel.cu:29-235 ; count/out must arrive zero-filled */
int vfi_oracle_flowproj_fwd(const float* flow, float* count, float* out,
                            int B, int H, int W, int fillhole);
/* A3b flowprojection_cuda_kernel.cu:237-301 */
int vfi_oracle_flowproj_bwd(const float* flow, const float* count, const float* gout,
                            float* gflow, int B, int H, int W);

/* A4  depthflowprojection_cuda_kernel.cu:29-241 */
int vfi_oracle_depthflowproj_fwd(const float* flow, const float* depth, float* count, float* out,
                                 int B, int H, int W, int fillhole, int fmad);
/* A4b depthflowprojection_cuda_kernel.cu:244-341 */
int vfi_oracle_depthflowproj_bwd(const float* flow, const float* depth, const float* count,
                                 const float* out, const float* gout, float* gflow, float* gdepth,
                                 int B, int H, int W);

/* MinDepthFlowProjection mindepthflowprojection_cuda_kernel.cu:27-206 (sequential raster order) / 209-331;
 * count/out/gflow must arrive zero-filled */
int vfi_oracle_mindepthflowproj_fwd(const float* flow, const float* weight, float* count, float* out,
                                    int B, int H, int W, int fillhole);
int vfi_oracle_mindepthflowproj_bwd(const float* flow, const float* weight, const float* count, const float* gout,
                                    float* gflow, int B, int H, int W);

/* A5  interpolation_cuda_kernel.cu:29-98 / 102-202 (InterpolationCh identical) */
int vfi_oracle_interp_fwd(const float* img, const float* flow, float* out,
                          int B, int C, int H, int W, int fmad);
int vfi_oracle_interp_bwd(const float* img, const float* flow, const float* gout,
                          float* gimg, float* gflow, int B, int C, int H, int W, int fmad);

/* A6  separableconv_cuda_kernel.cu:29-81 / 85-135 ; v,h,out are [B,*,H-fs+1,W-fs+1] */
int vfi_oracle_sepconv_fwd(const float* img, const float* v, const float* h, float* out,
                           int B, int C, int H, int W, int fs, int fmad);
int vfi_oracle_sepconv_bwd(const float* img, const float* v, const float* h, const float* gout,
                           float* gimg, float* gv, float* gh,
                           int B, int C, int H, int W, int fs);

/* A7  separableconvflow_cuda_kernel.cu:29-93 / 97-173 ; H,W are the image dims */
int vfi_oracle_sepconvflow_fwd(const float* v, const float* h, float* flow_out,
                               int B, int H, int W, int fs, int fmad);
int vfi_oracle_sepconvflow_bwd(const float* v, const float* h, const float* gflow,
                               float* gv, float* gh, int B, int H, int W, int fs, int fmad);

/* A8  correlation_cuda_kernel.cu:47-147 (+ correlation_cuda.cc:8-85 for sizes).
 * out = [B, (2*(md/s2)+1)^2, outH, outW]; order 0 = reference lane/tree order
 * (32 lanes, 16-8-4-2-1 shuffle tree), 1 = sequential over channels. */
int vfi_oracle_correlation_out_dims(int H, int W, int pad, int k, int md, int s1, int s2,
                                    int* outC, int* outH, int* outW);
int vfi_oracle_correlation_fwd(const float* f1, const float* f2, float* out,
                               int B, int C, int H, int W,
                               int pad, int k, int md, int s1, int s2, int order, int fmad);
/* A8b correlation_cuda_kernel.cu:151-334 */
int vfi_oracle_correlation_bwd(const float* f1, const float* f2, const float* gout,
                               float* g1, float* g2, int B, int C, int H, int W,
                               int pad, int k, int md, int s1, int s2);

/* ---- glue either side of the ops.  These restate torch built-ins the reference
 * calls (third-party: ATen of torch 1.0-1.4; restated from ATen's UpSampleBilinear2d and GridSampler
 * kernels) and are pinned against the torch CPU build of this image in tests/test_oracle.py. */

/* nn.Upsample(scale_factor=4, mode='bilinear') of (m0 * in) * m1, align_corners=False
 * (networks/DAIN_slowmotion.py:213-215) */
int vfi_oracle_flow_upsample4(const float* in, float* out, int B, int C, int hq, int wq,
                              float m0, float m1, int fmad);
/* PWCDCNet.warp (PWCNet/PWCNet.py:159-199) */
int vfi_oracle_pwc_warp(const float* x, const float* flo, float* out, int B, int C, int H, int W,
                        int align_corners, int fmad);

#ifdef __cplusplus
}
#endif
#endif
