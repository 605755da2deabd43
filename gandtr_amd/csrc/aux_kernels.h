// Launchers of the helper kernels in aux_kernels.hip (internal to the library; the public ABI is include/gandtr_hip.h).
// `f32` selects the activation element type: 0 = fp16 NHWC (default), 1 = fp32 NHWC ("f16x3" / "f16c" precision modes);
// gdt_k_pack_input also takes 2 = fp16 NHWC8 pixel words augmented with their own rounding residuals (conv_stem.hip, f16c form).
#pragma once
#include "gdt_common.h"

int gdt_k_pack_input(const float* x, void* y, int f32, int N, int C, int H, int W, int OH, int OW, float rscale, int resize,
                     const int* perm, const float* scale, const float* shift, hipStream_t st);
int gdt_in_stats_chunks(int HW);
int gdt_k_instance_norm(const void* x, const void* res, void* y, int f32, float* partial, float* mean_rstd, int N, int HW, int C,
                        float eps, int relu, hipStream_t st);
int gdt_k_instance_norm_fused(const void* x, const void* res, void* y, int f32, const float* tile_partials, int tiles_per_image,
                              int nphase, float* mean_rstd, int N, int HW, int C, float eps, int relu, hipStream_t st);
int gdt_k_instance_norm_stats(const void* x, int f32, int fused, float* partial, int tiles_per_image, int nphase, float* mean_rstd,
                              int N, int HW, int C, float eps, hipStream_t st);
int gdt_k_maxpool(const void* x, void* y, int f32, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, hipStream_t st);
int gdt_k_gem_l2n(const void* x, int f32, float* pooled, float* out, int N, int HW, int D, float p, float eps_gem, float eps_l2,
                  hipStream_t st);
int gdt_k_gem_l2n_nchw(const float* x, float* pooled, float* out, int N, int D, int HW, float p, float eps_gem, float eps_l2, hipStream_t st);
int gdt_k_l2n_rows(const float* x, float* y, int N, int D, float eps, hipStream_t st);
int gdt_k_ms_aggregate(const float* x, float* y, int S, int N, int D, float msp, hipStream_t st);
int gdt_k_whiten(const float* P, const float* m, const float* v, float* tmp, float* out, int N, int D, int dims,
                 hipStream_t st);
int gdt_k_whiten_f64(const double* P, const double* m, const double* v, double* tmp, double* out, int N, int D, int dims, hipStream_t st);
int gdt_k_unpack_output(const void* x, int f32, float* y, const float* bias, int N, int HW, int C, hipStream_t st);
int gdt_k_rowsplit_combine(const void* P, int f32, const float* bias, float* out, int N, int H, int W, int cp, int cout, int kw,
                           int pad, int reflect, int act, hipStream_t st);
int gdt_k_hed_score(const void* x, int f32, const float* w, float bias, float* score, long NP, int C, hipStream_t st);
int gdt_k_hed_fuse(const float* const* score, const int* h, const int* w, const float* fw, float fb, float* out, int N, int H,
                   int W, int sigmoid, hipStream_t st);
int gdt_conv_bn(int Cout);
