// HBM-bound helper kernels of the gandtr hot path (gfx950).  All activations are NHWC fp16; every lane moves 16 B.
//
//   pack_input      fp32 NCHW image -> fp16 NHWC8, optional bilinear resize (F.interpolate(scale_factor=s,
//                   mode='bilinear', align_corners=False), mdir/components/data/wrapper.py:225), channel permutation
//                   (RgbToBgrPre wrapper.py:351-364) and per-channel affine (MeanStdPost._adapt wrapper.py:172-175)
//   in_stats / in_finalize / in_apply
//                   nn.InstanceNorm2d(affine=False), eps 1e-5, biased variance (p2p_networks.py:29) + fused ReLU
//                   (:272) + fused residual add (ResnetBlock.forward :505)
//   maxpool         nn.MaxPool2d(2,2) (VGG16, HED hed.py:53) and (3,2,1) (ResNet-101 stem)
//   gem / l2n       cirtorch layers/functional.py:21-22, :130-131
//   ms_aggregate    CirMultiscaleAggregation.aggregate_tensor wrapper.py:236-245
//   whiten          CirtorchWhiten.postprocess wrapper.py:320-322
//   unpack_output   fp16 NHWC -> fp32 NCHW (feature taps, p2p_networks.py:316-334)
//   hed_score / hed_fuse   1x1 score convs, bilinear upsampling to the input size, 1x1 fusion, sigmoid (hed.py:67-83)
#include "gdt_common.h"
#include "aux_kernels.h"

namespace {

// activation element access: 8 consecutive channels as fp32, for fp16 (default) and fp32 ("f16x3" precision mode) tensors
__device__ __forceinline__ void load8(const f16* p, float (&v)[8]) {
    const f16x8 t = *(const f16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(f16* p, const float (&v)[8]) {
    f16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (f16)v[e];
    *(f16x8*)p = t;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------ pack_input
struct PackArgs {
    const float* x; void* y;
    int N, C, H, W, OH, OW;
    float rscale;          // 1 / scale_factor (torch's source-index scale when scale_factor is given)
    int resize;
    int aug;               // fp16 output only: slots [C, 2C) of the pixel word carry (v - fp16(v)) * 2^8 (conv_stem.hip, f16c form)
    int perm[8]; float scale[8], shift[8];
};

template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const PackArgs a) {
    const long total = (long)a.N * a.OH * a.OW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % a.OW);
        const long t = i / a.OW;
        const int oy = (int)(t % a.OH), n = (int)(t / a.OH);
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = 0.f;
        if (!a.resize) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < a.C) {
                    const float v = a.x[(((long)n * a.C + a.perm[c]) * a.H + oy) * a.W + ox];
                    o[c] = v * a.scale[c] + a.shift[c];
                }
        } else {
            // aten area_pixel_compute_source_index(align_corners=False): src = scale * (dst + 0.5) - 0.5, clamped at 0
            float sy = a.rscale * (oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
            float sx = a.rscale * (ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
            const float ly1 = sy - y0, ly0 = 1.f - ly1, lx1 = sx - x0, lx0 = 1.f - lx1;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < a.C) {
                    const float* p = a.x + ((long)n * a.C + a.perm[c]) * a.H * a.W;
                    const float v = ly0 * (lx0 * p[(long)y0 * a.W + x0] + lx1 * p[(long)y0 * a.W + x1]) +
                                    ly1 * (lx0 * p[(long)y1 * a.W + x0] + lx1 * p[(long)y1 * a.W + x1]);
                    o[c] = v * a.scale[c] + a.shift[c];
                }
        }
        if (a.aug) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < a.C) o[a.C + c] = (o[c] - (float)(f16)o[c]) * 256.f;
        }
        store8((T*)a.y + i * 8, o);
    }
}

// ------------------------------------------------------------------------------------------------ InstanceNorm
// stage 1: per (image, pixel chunk) partial sum / sum of squares for every channel.  partial[n][chunk][2][C]
template <typename T>
__global__ __launch_bounds__(256) void in_stats_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                       int HW, int C, int chunk_px) {
    __shared__ float red[256 * 16];
    const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int c8n = C >> 3;                       // 16-byte chunks per pixel
    const int c8 = threadIdx.x % c8n, prow = threadIdx.x / c8n, nprow = 256 / c8n;
    const int p0 = chunk * chunk_px, p1 = min(HW, p0 + chunk_px);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (prow < nprow) {
        const T* base = x + ((long)n * HW) * C + c8 * 8;
        for (int p = p0 + prow; p < p1; p += nprow) {
            float v[8];
            load8(base + (long)p * C, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x * 16 + e] = s[e]; red[threadIdx.x * 16 + 8 + e] = q[e]; }
    __syncthreads();
    // fixed-order reduction over pixel rows (deterministic)
    for (int o = threadIdx.x; o < C * 2; o += 256) {
        const int which = o / C, c = o % C;
        float acc = 0.f;
        for (int r = 0; r < nprow; ++r) acc += red[(r * c8n + (c >> 3)) * 16 + which * 8 + (c & 7)];
        partial[(((long)n * nchunks + chunk) * 2 + which) * C + c] = acc;
    }
}

// stage 2: mean / rstd per (n, c); fp64 accumulation over the chunk partials, biased variance.
__global__ __launch_bounds__(256) void in_finalize_kernel(const float* __restrict__ partial, float* __restrict__ mean_rstd,
                                                          int nchunks, int C, int HW, float eps, int NC) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NC) return;
    const int n = i / C, c = i % C;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < nchunks; ++k) {
        s += (double)partial[(((long)n * nchunks + k) * 2 + 0) * C + c];
        q += (double)partial[(((long)n * nchunks + k) * 2 + 1) * C + c];
    }
    const double m = s / HW;
    double var = q / HW - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean_rstd[(long)i * 2 + 0] = (float)m;
    mean_rstd[(long)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// stage 2 (fused-statistics variant): the partials were written by the conv epilogue, one [2][C] record per 128-row tile;
// tile index = phase * (N * tpi) + n * tpi + t   (phases: 1 for Conv2d, 4 for the ConvTranspose2d sub-pixel launches)
__global__ __launch_bounds__(1024) void in_finalize_tiles_kernel(const float* __restrict__ partial, float* __restrict__ mean_rstd,
                                                                 int tpi, int nphase, int N, int C, int HW, float eps) {
    // grid (C / 64, N): 64 channels x 16 record lanes per workgroup; records are strided over the lanes, four independent
    // loads in flight per lane (one at a time, a 256x256 layer's 512 records per image took 69 us of pure load latency), and
    // everything is merged in a fixed order (fp64), so the result does not depend on scheduling
    constexpr int RL = 16;
    __shared__ double red[2][RL][64];
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), lanegrp = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const int nrec = nphase * tpi;
        auto rec = [&](int r, float& a, float& b) {
            if (r < nrec) {
                const int ph = r / tpi, t = r - ph * tpi;
                const long tile = (long)ph * N * tpi + (long)n * tpi + t;
                a = partial[(tile * 2 + 0) * C + c]; b = partial[(tile * 2 + 1) * C + c];
            } else { a = 0.f; b = 0.f; }
        };
        for (int r = lanegrp; r < nrec; r += 4 * RL) {
            float a0, b0, a1, b1, a2, b2, a3, b3;
            rec(r, a0, b0); rec(r + RL, a1, b1); rec(r + 2 * RL, a2, b2); rec(r + 3 * RL, a3, b3);
            s += (double)a0; q += (double)b0; s += (double)a1; q += (double)b1;
            s += (double)a2; q += (double)b2; s += (double)a3; q += (double)b3;
        }
    }
    red[0][lanegrp][threadIdx.x & 63] = s; red[1][lanegrp][threadIdx.x & 63] = q;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        s = 0.0; q = 0.0;
#pragma unroll
        for (int l = 0; l < RL; ++l) { s += red[0][l][threadIdx.x]; q += red[1][l][threadIdx.x]; }
        const double m = s / HW;
        double var = q / HW - m * m;
        var = var < 0.0 ? 0.0 : var;
        mean_rstd[((long)n * C + c) * 2 + 0] = (float)m;
        mean_rstd[((long)n * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// stage 3: y = relu?((x - mean) * rstd) (+ residual)
template <typename T>
__global__ __launch_bounds__(256) void in_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean_rstd,
                                                       const T* __restrict__ res, T* __restrict__ y,
                                                       long HW, int C, int relu, long total8) {
    const int c8n = C >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        const long n = (i / c8n) / HW;
        const float* mr = mean_rstd + (n * C + c8 * 8) * 2;
        float v[8], r[8];
        load8(x + i * 8, v);
        if (res) load8(res + i * 8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (v[e] - mr[2 * e]) * mr[2 * e + 1];
            if (relu) f = fmaxf(f, 0.f);
            if (res) f += r[e];
            v[e] = f;
        }
        store8(y + i * 8, v);
    }
}

// ------------------------------------------------------------------------------------------------ max pooling
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                      int C, int OH, int OW, int k, int s, int p) {
    const int c8n = C >> 3;
    const long total = (long)N * OH * OW * c8n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        long t = i / c8n;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH), n = (int)(t / OH);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * s - p + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * s - p + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                float v[8];
                load8(x + (((long)n * H + iy) * W + ix) * C + c8 * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        store8(y + i * 8, m);
    }
}

// ------------------------------------------------------------------------------------------------ GeM + L2N
// grid (D/64, N); 8 lanes cover 64 channels (16 B each), 32 pixel rows per pass; fixed-order LDS reduction.
template <typename T>
__global__ __launch_bounds__(256) void gem_kernel(const T* __restrict__ x, float* __restrict__ pooled, int HW, int D,
                                                  float p, float eps) {
    __shared__ float red[32][64];
    const int n = blockIdx.y, d0 = blockIdx.x * 64;
    const int c8 = threadIdx.x & 7, prow = threadIdx.x >> 3;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const T* base = x + (long)n * HW * D + d0 + c8 * 8;
    const bool cube = (p == 3.0f);
    for (int px = prow; px < HW; px += 32) {
        float v[8];
        load8(base + (long)px * D, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = fmaxf(v[e], eps);
            s[e] += cube ? f * f * f : powf(f, p);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[prow][c8 * 8 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float acc = 0.f;
        for (int r = 0; r < 32; ++r) acc += red[r][threadIdx.x];
        pooled[(long)n * D + d0 + threadIdx.x] = powf(acc / HW, 1.0f / p);
    }
}

// GeM over NCHW fp32 feature maps (the reference layout, cirtorch layers/functional.py:21-22): one wavefront per (n, c) plane, lanes
// stride over the H*W contiguous pixels, butterfly reduction; pooled[n][c] = (mean clamp(x, eps)^p)^(1/p)
__global__ __launch_bounds__(256) void gem_nchw_kernel(const float* __restrict__ x, float* __restrict__ pooled, long planes, int HW, float p, float eps) {
    const long plane = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float* xp = x + plane * HW;
    const bool cube = (p == 3.0f);
    float s = 0.f;
    for (int i = threadIdx.x & 63; i < HW; i += 64) {
        const float f = fmaxf(xp[i], eps);
        s += cube ? f * f * f : powf(f, p);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) pooled[plane] = powf(s / HW, 1.0f / p);
}

// y[n][:] = x[n][:] / (||x[n]||_2 + eps); one workgroup per row
__global__ __launch_bounds__(256) void l2n_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int D, float eps) {
    __shared__ float red[4];
    const float* xr = x + (long)blockIdx.x * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) s += xr[i] * xr[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]) + eps;
    for (int i = threadIdx.x; i < D; i += 256) y[(long)blockIdx.x * D + i] = xr[i] / nrm;
}

// v[n][d] = (mean_s x[s][n][d]^msp)^(1/msp); then v /= ||v|| (no eps)
__global__ __launch_bounds__(256) void ms_aggregate_kernel(const float* __restrict__ x, float* __restrict__ y, int S, int N,
                                                           int D, float msp) {
    __shared__ float red[4];
    extern __shared__ float vbuf[];
    const int n = blockIdx.x;
    float ss = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += powf(x[((long)s * N + n) * D + i], msp);
        const float v = powf(acc / S, 1.0f / msp);
        vbuf[i] = v;
        ss += v * v;
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    for (int i = threadIdx.x; i < D; i += 256) y[(long)n * D + i] = vbuf[i] / nrm;
}

// X[n][r] = sum_d P[r][d] * (v[n][d] - m[d]) for r < dims; one wavefront per (row, image)
__global__ __launch_bounds__(256) void whiten_matvec_kernel(const float* __restrict__ P, const float* __restrict__ m,
                                                            const float* __restrict__ v, float* __restrict__ X, int D, int dims) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y, lane = threadIdx.x & 63;
    if (r >= dims) return;
    const float* pr = P + (long)r * D;
    const float* vr = v + (long)n * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += pr[d] * (vr[d] - m[d]);
    s = wave_sum(s);
    if (lane == 0) X[(long)n * dims + r] = s;
}

// float64 form of the two kernels above: the reference's `whiten` STAGE applies the learned whitening with numpy in float64
// (whitenapply, cirtorch/utils/whiten.py:4-12: P is float64, so X - m and the product are promoted)
__global__ __launch_bounds__(256) void whiten_matvec_f64_kernel(const double* __restrict__ P, const double* __restrict__ m,
                                                                const double* __restrict__ v, double* __restrict__ X, int D, int dims) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y, lane = threadIdx.x & 63;
    if (r >= dims) return;
    const double* pr = P + (long)r * D;
    const double* vr = v + (long)n * D;
    double s = 0.0;
    for (int d = lane; d < D; d += 64) s += pr[d] * (vr[d] - m[d]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) X[(long)n * dims + r] = s;
}
__global__ __launch_bounds__(256) void l2n_rows_f64_kernel(const double* __restrict__ x, double* __restrict__ y, int D, double eps) {
    __shared__ double red[4];
    const double* xr = x + (long)blockIdx.x * D;
    double s = 0.0;
    for (int i = threadIdx.x; i < D; i += 256) s += xr[i] * xr[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]) + eps;
    for (int i = threadIdx.x; i < D; i += 256) y[(long)blockIdx.x * D + i] = xr[i] / nrm;
}

// ------------------------------------------------------------------------------------------------ taps / outputs
template <typename T>
__global__ __launch_bounds__(256) void unpack_output_kernel(const T* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ bias, int N, int HW, int C) {
    const long total = (long)N * C * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const long t = i / HW;
        const int c = (int)(t % C), n = (int)(t / C);
        y[i] = (float)x[((long)n * HW + p) * C + c] + (bias ? bias[c] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ row-split conv head
// Second half of a k x k conv with very few output channels (generator head 64 -> 3, k = 7, p2p_networks.py:309-311).  The
// implicit GEMM computed P[n][y][x'][kx*cout + co] = sum_{ky,c} in[y+ky-pad][x'][c] * W[co][c][ky][kx] (a k x 1 conv with
// k*cout output channels, so the MFMA N tile is 21/32 used instead of 3/32); this kernel adds the k horizontally shifted
// partials, the bias and the activation and writes fp32 NCHW.
template <typename T>
__global__ __launch_bounds__(256) void rowsplit_combine_kernel(const T* __restrict__ P, const float* __restrict__ bias,
                                                               float* __restrict__ out, int N, int H, int W, int cp, int cout,
                                                               int kw, int pad, int reflect, int act) {
    const long total = (long)N * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        const long row = i / W;                       // n * H + y
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int kx = 0; kx < kw; ++kx) {
            int xs = x + kx - pad;
            if (reflect) xs = xs < 0 ? -xs : (xs >= W ? 2 * W - 2 - xs : xs);
            else if ((unsigned)xs >= (unsigned)W) continue;
            const T* p = P + (row * W + xs) * cp + kx * cout;
            for (int co = 0; co < cout; ++co) acc[co] += (float)p[co];
        }
        const long n = row / H; const int y = (int)(row % H);
        for (int co = 0; co < cout; ++co) {
            float v = acc[co] + (bias ? bias[co] : 0.f);
            if (act == 1) v = tanhf(v);
            else if (act == 2) v = 1.f / (1.f + __expf(-v));
            out[((n * cout + co) * H + y) * W + x] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ HED head
// score[n][p] = b + sum_c x[n][p][c] * w[c]; one wavefront per pixel group of 8 (8 lanes per pixel)
template <typename T>
__global__ __launch_bounds__(256) void hed_score_kernel(const T* __restrict__ x, const float* __restrict__ w, float bias,
                                                        float* __restrict__ score, long NP, int C) {
    const long p = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int sub = threadIdx.x & 7;
    float s = 0.f;
    if (p < NP) {
        const T* xr = x + p * C;
        for (int c = sub * 8; c < C; c += 64) {
            float v[8];
            load8(xr + c, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e] * w[c + e];
        }
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (p < NP && sub == 0) score[p] = s + bias;
}

struct HedFuseArgs {
    const float* score[5]; int h[5], w[5];
    float fw[5], fb;
    float* out; int N, H, W, sigmoid;
};

// F.interpolate(size=(H,W), mode='bilinear', align_corners=False) of each score map (scale = in/out), 1x1 fusion, sigmoid
__global__ __launch_bounds__(256) void hed_fuse_kernel(const HedFuseArgs a) {
    const long total = (long)a.N * a.H * a.W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % a.W);
        const long t = i / a.W;
        const int oy = (int)(t % a.H), n = (int)(t / a.H);
        float acc = a.fb;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int h = a.h[k], w = a.w[k];
            const float ry = (float)h / a.H, rx = (float)w / a.W;
            float sy = ry * (oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
            float sx = rx * (ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float ly1 = sy - y0, ly0 = 1.f - ly1, lx1 = sx - x0, lx0 = 1.f - lx1;
            const float* p = a.score[k] + (long)n * h * w;
            const float v = ly0 * (lx0 * p[y0 * w + x0] + lx1 * p[y0 * w + x1]) + ly1 * (lx0 * p[y1 * w + x0] + lx1 * p[y1 * w + x1]);
            acc += a.fw[k] * v;
        }
        a.out[i] = a.sigmoid ? 1.f / (1.f + expf(-acc)) : acc;
    }
}

inline int grid_for(long work_items, int cap = 256 * 16) {
    long g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// ================================================================================================ launchers
// `f32` selects the activation element type: 0 = fp16 (default), 1 = fp32 ("f16x3" precision mode).
#define LAUNCH_T(kernel, f32, grid, lds, st, ...)                                                     \
    do {                                                                                              \
        if (f32) hipLaunchKernelGGL(kernel<float>, grid, dim3(256), lds, st, __VA_ARGS__);            \
        else hipLaunchKernelGGL(kernel<f16>, grid, dim3(256), lds, st, __VA_ARGS__);                  \
        GDT_CHECK_HIP(hipGetLastError());                                                             \
    } while (0)
#define TP(T, p) ((T*)(p))

int gdt_k_pack_input(const float* x, void* y, int f32, int N, int C, int H, int W, int OH, int OW, float rscale, int resize,
                     const int* perm, const float* scale, const float* shift, hipStream_t st) {
    GDT_REQUIRE(C >= 1 && C <= 8, "pack_input supports 1..8 channels");
    PackArgs a;
    a.x = x; a.y = y; a.N = N; a.C = C; a.H = H; a.W = W; a.OH = OH; a.OW = OW; a.rscale = rscale; a.resize = resize;
    for (int c = 0; c < 8; ++c) {
        a.perm[c] = (perm && c < C) ? perm[c] : (c < C ? c : 0);
        a.scale[c] = (scale && c < C) ? scale[c] : 1.f;
        a.shift[c] = (shift && c < C) ? shift[c] : 0.f;
    }
    a.aug = f32 == 2 ? 1 : 0;
    GDT_REQUIRE(!a.aug || 2 * C <= 8, "augmented pixel words need 2 * channels <= 8");
    LAUNCH_T(pack_input_kernel, f32 == 1, dim3(grid_for((long)N * OH * OW)), 0, st, a);
    return GDT_OK;
}

int gdt_in_stats_chunks(int HW) { int c = (HW + 1023) / 1024; return c < 1 ? 1 : c; }

static int launch_apply(const void* x, const void* res, void* y, int f32, const float* mean_rstd, int N, int HW, int C, int relu,
                        hipStream_t st) {
    const long total8 = (long)N * HW * (C / 8);
    if (f32)
        hipLaunchKernelGGL(in_apply_kernel<float>, dim3(grid_for(total8)), dim3(256), 0, st, (const float*)x, mean_rstd,
                           (const float*)res, (float*)y, (long)HW, C, relu, total8);
    else
        hipLaunchKernelGGL(in_apply_kernel<f16>, dim3(grid_for(total8)), dim3(256), 0, st, (const f16*)x, mean_rstd, (const f16*)res,
                           (f16*)y, (long)HW, C, relu, total8);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_instance_norm(const void* x, const void* res, void* y, int f32, float* partial, float* mean_rstd, int N, int HW, int C,
                        float eps, int relu, hipStream_t st) {
    GDT_REQUIRE(C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, "InstanceNorm needs a power-of-two channel count <= 2048");
    const int nchunks = gdt_in_stats_chunks(HW);
    const int chunk_px = (HW + nchunks - 1) / nchunks;
    if (f32) hipLaunchKernelGGL(in_stats_kernel<float>, dim3(nchunks, N), dim3(256), 0, st, (const float*)x, partial, HW, C, chunk_px);
    else hipLaunchKernelGGL(in_stats_kernel<f16>, dim3(nchunks, N), dim3(256), 0, st, (const f16*)x, partial, HW, C, chunk_px);
    GDT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(in_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, (const float*)partial, mean_rstd, nchunks, C,
                       HW, eps, N * C);
    GDT_CHECK_HIP(hipGetLastError());
    return launch_apply(x, res, y, f32, mean_rstd, N, HW, C, relu, st);
}

int gdt_k_instance_norm_fused(const void* x, const void* res, void* y, int f32, const float* tile_partials, int tiles_per_image,
                              int nphase, float* mean_rstd, int N, int HW, int C, float eps, int relu, hipStream_t st) {
    hipLaunchKernelGGL(in_finalize_tiles_kernel, dim3((C + 63) / 64, N), dim3(1024), 0, st, tile_partials, mean_rstd,
                       tiles_per_image, nphase, N, C, HW, eps);
    GDT_CHECK_HIP(hipGetLastError());
    return launch_apply(x, res, y, f32, mean_rstd, N, HW, C, relu, st);
}

// mean / rstd only (the normalisation itself is applied by the consuming conv's input staging)
int gdt_k_instance_norm_stats(const void* x, int f32, int fused, float* partial, int tiles_per_image, int nphase, float* mean_rstd,
                              int N, int HW, int C, float eps, hipStream_t st) {
    if (fused) {
        hipLaunchKernelGGL(in_finalize_tiles_kernel, dim3((C + 63) / 64, N), dim3(1024), 0, st, (const float*)partial, mean_rstd,
                           tiles_per_image, nphase, N, C, HW, eps);
        GDT_CHECK_HIP(hipGetLastError());
        return GDT_OK;
    }
    GDT_REQUIRE(C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, "InstanceNorm needs a power-of-two channel count <= 2048");
    const int nchunks = gdt_in_stats_chunks(HW);
    const int chunk_px = (HW + nchunks - 1) / nchunks;
    if (f32) hipLaunchKernelGGL(in_stats_kernel<float>, dim3(nchunks, N), dim3(256), 0, st, (const float*)x, partial, HW, C, chunk_px);
    else hipLaunchKernelGGL(in_stats_kernel<f16>, dim3(nchunks, N), dim3(256), 0, st, (const f16*)x, partial, HW, C, chunk_px);
    GDT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(in_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, (const float*)partial, mean_rstd, nchunks, C,
                       HW, eps, N * C);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_maxpool(const void* x, void* y, int f32, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, hipStream_t st) {
    GDT_REQUIRE(C % 8 == 0, "maxpool needs C % 8 == 0");
    const dim3 grid(grid_for((long)N * OH * OW * (C / 8)));
    if (f32) hipLaunchKernelGGL(maxpool_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, OH, OW, k, s, p);
    else hipLaunchKernelGGL(maxpool_kernel<f16>, grid, dim3(256), 0, st, (const f16*)x, (f16*)y, N, H, W, C, OH, OW, k, s, p);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_gem_l2n(const void* x, int f32, float* pooled, float* out, int N, int HW, int D, float p, float eps_gem, float eps_l2,
                  hipStream_t st) {
    GDT_REQUIRE(D % 64 == 0, "GeM needs D % 64 == 0");
    if (f32) hipLaunchKernelGGL(gem_kernel<float>, dim3(D / 64, N), dim3(256), 0, st, (const float*)x, pooled, HW, D, p, eps_gem);
    else hipLaunchKernelGGL(gem_kernel<f16>, dim3(D / 64, N), dim3(256), 0, st, (const f16*)x, pooled, HW, D, p, eps_gem);
    GDT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(l2n_rows_kernel, dim3(N), dim3(256), 0, st, (const float*)pooled, out, D, eps_l2);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_gem_l2n_nchw(const float* x, float* pooled, float* out, int N, int D, int HW, float p, float eps_gem, float eps_l2, hipStream_t st) {
    const long planes = (long)N * D;
    hipLaunchKernelGGL(gem_nchw_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, st, x, pooled, planes, HW, p, eps_gem);
    GDT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(l2n_rows_kernel, dim3(N), dim3(256), 0, st, (const float*)pooled, out, D, eps_l2);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_l2n_rows(const float* x, float* y, int N, int D, float eps, hipStream_t st) {
    hipLaunchKernelGGL(l2n_rows_kernel, dim3(N), dim3(256), 0, st, x, y, D, eps);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_ms_aggregate(const float* x, float* y, int S, int N, int D, float msp, hipStream_t st) {
    GDT_REQUIRE(D * sizeof(float) <= 48 * 1024, "ms_aggregate: D too large");
    hipLaunchKernelGGL(ms_aggregate_kernel, dim3(N), dim3(256), D * sizeof(float), st, x, y, S, N, D, msp);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_whiten(const float* P, const float* m, const float* v, float* tmp, float* out, int N, int D, int dims,
                 hipStream_t st) {
    hipLaunchKernelGGL(whiten_matvec_kernel, dim3((dims + 3) / 4, N), dim3(256), 0, st, P, m, v, tmp, D, dims);
    GDT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(l2n_rows_kernel, dim3(N), dim3(256), 0, st, (const float*)tmp, out, dims, 1e-6f);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_whiten_f64(const double* P, const double* m, const double* v, double* tmp, double* out, int N, int D, int dims, hipStream_t st) {
    hipLaunchKernelGGL(whiten_matvec_f64_kernel, dim3((dims + 3) / 4, N), dim3(256), 0, st, P, m, v, tmp, D, dims);
    GDT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(l2n_rows_f64_kernel, dim3(N), dim3(256), 0, st, (const double*)tmp, out, dims, 1e-6);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_unpack_output(const void* x, int f32, float* y, const float* bias, int N, int HW, int C, hipStream_t st) {
    const dim3 grid(grid_for((long)N * HW * C));
    if (f32) hipLaunchKernelGGL(unpack_output_kernel<float>, grid, dim3(256), 0, st, (const float*)x, y, bias, N, HW, C);
    else hipLaunchKernelGGL(unpack_output_kernel<f16>, grid, dim3(256), 0, st, (const f16*)x, y, bias, N, HW, C);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_rowsplit_combine(const void* P, int f32, const float* bias, float* out, int N, int H, int W, int cp, int cout, int kw,
                           int pad, int reflect, int act, hipStream_t st) {
    GDT_REQUIRE(cout >= 1 && cout <= 4, "row-split head supports up to 4 output channels");
    const dim3 grid(grid_for((long)N * H * W));
    if (f32) hipLaunchKernelGGL(rowsplit_combine_kernel<float>, grid, dim3(256), 0, st, (const float*)P, bias, out, N, H, W, cp, cout, kw, pad, reflect, act);
    else hipLaunchKernelGGL(rowsplit_combine_kernel<f16>, grid, dim3(256), 0, st, (const f16*)P, bias, out, N, H, W, cp, cout, kw, pad, reflect, act);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_hed_score(const void* x, int f32, const float* w, float bias, float* score, long NP, int C, hipStream_t st) {
    GDT_REQUIRE(C % 8 == 0, "hed_score needs C % 8 == 0");
    const dim3 grid((int)((NP + 31) / 32));
    if (f32) hipLaunchKernelGGL(hed_score_kernel<float>, grid, dim3(256), 0, st, (const float*)x, w, bias, score, NP, C);
    else hipLaunchKernelGGL(hed_score_kernel<f16>, grid, dim3(256), 0, st, (const f16*)x, w, bias, score, NP, C);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_k_hed_fuse(const float* const* score, const int* h, const int* w, const float* fw, float fb, float* out, int N, int H,
                   int W, int sigmoid, hipStream_t st) {
    HedFuseArgs a;
    for (int k = 0; k < 5; ++k) { a.score[k] = score[k]; a.h[k] = h[k]; a.w[k] = w[k]; a.fw[k] = fw[k]; }
    a.fb = fb; a.out = out; a.N = N; a.H = H; a.W = W; a.sigmoid = sigmoid;
    hipLaunchKernelGGL(hed_fuse_kernel, dim3(grid_for((long)N * H * W)), dim3(256), 0, st, a);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

// ------------------------------------------------------------------------------------------------ measurement aid
namespace {
__global__ __launch_bounds__(512) void mfma_only_kernel(const f16* __restrict__ src, float* __restrict__ out, int iters) {
    f16x8 a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const f16x8*)(src + (threadIdx.x * 4 + i) * 8);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = *(const f16x8*)(src + 16384 + (threadIdx.x * 2 + i) * 8);
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i & 1], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 1234.5f) out[0] = s;
}
}  // namespace

extern "C" int gdt_mfma_only_tflops(int millis, double* tflops, void* stream) {
    GDT_REQUIRE(tflops && millis > 0 && millis <= 1000, "gdt_mfma_only_tflops(millis in 1..1000, tflops)");
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 0;
    GDT_CHECK_HIP(hipGetDevice(&dev));
    GDT_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    f16* src = nullptr; float* out = nullptr;
    GDT_CHECK_HIP(hipMalloc((void**)&src, 65536 * sizeof(f16)));
    GDT_CHECK_HIP(hipMalloc((void**)&out, 64));
    std::vector<f16> h(65536);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (f16)(((x >> 8) & 0xffff) / 65536.f - 0.5f); }
    GDT_CHECK_HIP(hipMemcpyAsync(src, h.data(), h.size() * sizeof(f16), hipMemcpyHostToDevice, st));
    hipEvent_t e0, e1;
    GDT_CHECK_HIP(hipEventCreate(&e0)); GDT_CHECK_HIP(hipEventCreate(&e1));
    // 8 MFMAs x 32 cycles x 2 waves per SIMD per iteration: ~20 000 iterations are ~7 ms at 1.5 GHz; warm up, then time
    const int per_launch = 20000, launches = millis / 7 + 1;
    hipLaunchKernelGGL(mfma_only_kernel, dim3(cus), dim3(512), 0, st, src, out, per_launch);
    GDT_CHECK_HIP(hipEventRecord(e0, st));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_only_kernel, dim3(cus), dim3(512), 0, st, src, out, per_launch);
    GDT_CHECK_HIP(hipEventRecord(e1, st));
    GDT_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GDT_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *tflops = (double)cus * 8 * per_launch * 8 * 2.0 * 32 * 32 * 16 * launches / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(out);
    return GDT_OK;
}
