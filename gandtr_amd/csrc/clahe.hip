// CLAHE post-processing on the device: the step between the generator and the embedder in the
// augment -> embed chain, which the reference runs per image as device -> CPU -> cv2 -> device.
//   ClahePost.postprocess          mdir/components/data/wrapper.py:325-348
//   ImageClahe.apply               mdir/components/data/transform/functional.py:151-158 (apply_lightness_transform :81-85)
//   rgb2normspace / normspace2rgb  functional.py:28-36 / :55-63 ("lab")
//   ChannelClahe.apply_clahe       functional.py:147-148 ((chan * 255).astype(uint8) -> cv2.createCLAHE(...).apply -> / 255)
// The arithmetic is OpenCV's (cv::CLAHE for 8-bit planes, float32 COLOR_RGB2LAB / COLOR_LAB2RGB); oracle/clahe_oracle.py holds
// the restatement and says what is pinned.  HBM-bound byte / float work, three launches per batch:
//   1. clahe_hist_kernel:  one workgroup per (image, tile, row part): L of every pixel of the REFLECT_101-extended tile
//      (sRGB curve + the Y row of the XYZ matrix + one cube root), 8-bit quantised, written to a 1 B/pixel plane and counted in
//      four wave-private LDS histograms (integer atomics: order-independent), merged into the tile's global histogram.
//   2. clahe_lut_kernel:   one workgroup per (image, tile), one lane per bin: clip, redistribute (batch + strided residual),
//      block scan, LUT = rint(cumsum * 255 / area).
//   3. clahe_apply_kernel: four pixels of a row per lane: RGB -> Lab (a, b), L from the plane through the bilinear blend of the
//      four neighbouring tiles' LUTs (unfused float32, as OpenCV evaluates it), Lab -> RGB, output affine; 16-byte loads / stores.
// Algorithmic bytes per pixel: 12 (read RGB) + 1 (write L) in pass 1, 12 + 1 + 12 in pass 3 = 38 B.
#include <string.h>

#include "../../include/gandtr_hip.h"
#include "gdt_common.h"

// OpenCV evaluates the LUT blend (and numpy the oracle) with separately rounded float32 multiplies and adds.  hipcc's default
// -ffp-contract=fast would fuse them -- also through HIP's __fmul_rn / __fadd_rn, which are plain operators compiled under that
// default -- so contraction is switched off for this file and the exact steps use the local helpers below.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }

struct ClaheGeom {
    int n, h, w, tiles_x, tiles_y, th, tw;      // tile size in pixels of the extended image
    int limit;                                  // integer clip limit per bin (0: no clipping)
    float lut_scale, inv_tw, inv_th;
    int rows_per_part, parts;                   // histogram pass: tile rows per workgroup, workgroups per tile
};

struct LabParams {
    float in_scale[3], in_shift[3];             // rgb = x * in_scale + in_shift
    float out_mean[3], out_std[3];              // y = (rgb - out_mean) / out_std
    float fwd[9], inv[9];                       // RGB -> XYZ / white, XYZ * white -> RGB (row-major)
};

__device__ __forceinline__ float fast_pow(float x, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x)); }

__device__ __forceinline__ float srgb_to_linear(float c) {
    c = fminf(fmaxf(c, 0.f), 1.f);
    return c <= 0.04045f ? c * (1.0f / 12.92f) : fast_pow((c + 0.055f) * (1.0f / 1.055f), 2.4f);
}
__device__ __forceinline__ float linear_to_srgb(float c) {
    c = fminf(fmaxf(c, 0.f), 1.f);
    return c <= 0.0031308f ? c * 12.92f : fast_pow(c, 1.0f / 2.4f) * 1.055f - 0.055f;
}
__device__ __forceinline__ float lab_f(float v) { return v > 0.008856f ? fast_pow(v, 1.0f / 3.0f) : 7.787f * v + (16.0f / 116.0f); }

// 8-bit lightness of one pixel: (L / 100 * 255) truncated, functional.py:148
__device__ __forceinline__ unsigned lightness_u8(float r, float g, float b, const LabParams& p) {
    const float lr = srgb_to_linear(add_rn(mul_rn(r, p.in_scale[0]), p.in_shift[0]));
    const float lg = srgb_to_linear(add_rn(mul_rn(g, p.in_scale[1]), p.in_shift[1]));
    const float lb = srgb_to_linear(add_rn(mul_rn(b, p.in_scale[2]), p.in_shift[2]));
    const float y = add_rn(add_rn(mul_rn(p.fwd[3], lr), mul_rn(p.fwd[4], lg)), mul_rn(p.fwd[5], lb));
    const float L = y > 0.008856f ? sub_rn(mul_rn(116.0f, lab_f(y)), 16.0f) : mul_rn(903.3f, y);
    const float q = mul_rn(div_rn(L, 100.0f), 255.0f);
    return (unsigned)min(max((int)q, 0), 255);
}

template <bool RGB>
__global__ __launch_bounds__(256) void clahe_hist_kernel(ClaheGeom g, const unsigned char* __restrict__ src8, const float* __restrict__ x,
                                                         LabParams p, unsigned char* __restrict__ lplane, unsigned* __restrict__ hist) {
    __shared__ unsigned sh[4][256];
    const int tid = threadIdx.x, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k][tid] = 0;
    __syncthreads();
    const int tiles = g.tiles_x * g.tiles_y;
    const int part = blockIdx.x % g.parts, tile = (blockIdx.x / g.parts) % tiles, img = blockIdx.x / (g.parts * tiles);
    const int ty = tile / g.tiles_x, tx = tile % g.tiles_x;
    const int r0 = part * g.rows_per_part, r1 = min(r0 + g.rows_per_part, g.th);
    const int count = (r1 - r0) * g.tw;
    const long plane = (long)g.h * g.w;
    for (int i = tid; i < count; i += 256) {
        const int r = i / g.tw, c = i - r * g.tw;
        const int ey = ty * g.th + r0 + r, ex = tx * g.tw + c;
        const int sy = ey < g.h ? ey : 2 * (g.h - 1) - ey, sx = ex < g.w ? ex : 2 * (g.w - 1) - ex;      // BORDER_REFLECT_101
        const long off = (long)sy * g.w + sx;
        unsigned v;
        if (RGB) {
            const float* px = x + (long)img * 3 * plane + off;
            v = lightness_u8(px[0], px[plane], px[2 * plane], p);
            if (ey < g.h && ex < g.w) lplane[(long)img * plane + off] = (unsigned char)v;
        } else {
            v = src8[(long)img * plane + off];
        }
        atomicAdd(&sh[wave][v], 1u);
    }
    __syncthreads();
    const unsigned s = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
    if (s) atomicAdd(&hist[((long)img * tiles + tile) * 256 + tid], s);
}

// one workgroup per (image, tile); lane = histogram bin (CLAHE_CalcLut_Body of OpenCV's clahe.cpp)
__global__ __launch_bounds__(256) void clahe_lut_kernel(ClaheGeom g, const unsigned* __restrict__ hist, unsigned char* __restrict__ lut) {
    __shared__ int red[4];
    __shared__ int scan[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int hv = (int)hist[(long)blockIdx.x * 256 + tid];
    if (g.limit > 0) {
        int excess = max(hv - g.limit, 0);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) excess += __shfl_xor(excess, m);
        if (lane == 0) red[wave] = excess;
        __syncthreads();
        const int clipped = red[0] + red[1] + red[2] + red[3];
        const int batch = clipped / 256, residual = clipped - batch * 256;
        hv = min(hv, g.limit) + batch;
        if (residual) {
            const int step = max(256 / residual, 1);
            if (tid % step == 0 && tid / step < residual) ++hv;
        }
    }
    int s = hv;                                 // inclusive scan over the 256 bins
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int t = __shfl_up(s, m);
        if (lane >= m) s += t;
    }
    if (lane == 63) scan[wave] = s;
    __syncthreads();
    for (int k = 0; k < wave; ++k) s += scan[k];
    const float f = __builtin_rintf(mul_rn((float)s, g.lut_scale));
    lut[(long)blockIdx.x * 256 + tid] = (unsigned char)fminf(fmaxf(f, 0.f), 255.f);
}

struct Blend { int i1, i2; float a, a1; };
__device__ __forceinline__ Blend blend_of(int pos, float inv_t, int tiles) {
    const float f = sub_rn(mul_rn((float)pos, inv_t), 0.5f);
    const int t1 = (int)floorf(f);
    Blend b;
    b.a = sub_rn(f, (float)t1);
    b.a1 = sub_rn(1.0f, b.a);
    b.i1 = max(t1, 0);
    b.i2 = min(t1 + 1, tiles - 1);
    return b;
}

// CLAHE_Interpolation_Body: unfused float32, round half to even, saturate
__device__ __forceinline__ unsigned blend_lut(const unsigned char* __restrict__ l1, const unsigned char* __restrict__ l2, const Blend& bx,
                                              const Blend& by, unsigned v) {
    const float v11 = (float)l1[bx.i1 * 256 + v], v12 = (float)l1[bx.i2 * 256 + v];
    const float v21 = (float)l2[bx.i1 * 256 + v], v22 = (float)l2[bx.i2 * 256 + v];
    const float top = add_rn(mul_rn(v11, bx.a1), mul_rn(v12, bx.a));
    const float bot = add_rn(mul_rn(v21, bx.a1), mul_rn(v22, bx.a));
    const float res = __builtin_rintf(add_rn(mul_rn(top, by.a1), mul_rn(bot, by.a)));
    return (unsigned)fminf(fmaxf(res, 0.f), 255.f);
}

// 8-bit plane in, 8-bit plane out; VEC pixels of one row per lane
template <int VEC>
__global__ __launch_bounds__(256) void clahe_apply_u8_kernel(ClaheGeom g, const unsigned char* __restrict__ src, const unsigned char* __restrict__ lut,
                                                             unsigned char* __restrict__ dst) {
    const int wq = (g.w + VEC - 1) / VEC;
    const long total = (long)g.n * g.h * wq;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int xq = (int)(id % wq);
    const long row = id / wq;
    const int y = (int)(row % g.h), img = (int)(row / g.h);
    const Blend by = blend_of(y, g.inv_th, g.tiles_y);
    const unsigned char* limg = lut + (long)img * g.tiles_x * g.tiles_y * 256;
    const unsigned char* l1 = limg + (long)by.i1 * g.tiles_x * 256;
    const unsigned char* l2 = limg + (long)by.i2 * g.tiles_x * 256;
    const long base = row * g.w + (long)xq * VEC;
    unsigned char v[VEC];
    if (VEC == 4) *(uchar4*)v = *(const uchar4*)(src + base);
    else v[0] = src[base];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = (unsigned char)blend_lut(l1, l2, blend_of(xq * VEC + k, g.inv_tw, g.tiles_x), by, v[k]);
    if (VEC == 4) *(uchar4*)(dst + base) = *(uchar4*)v;
    else dst[base] = v[0];
}

// fp32 NCHW RGB in / out with the lightness plane of pass 1
template <int VEC>
__global__ __launch_bounds__(256) void clahe_apply_lab_kernel(ClaheGeom g, const float* __restrict__ x, const unsigned char* __restrict__ lplane,
                                                              const unsigned char* __restrict__ lut, LabParams p, float* __restrict__ out) {
    const int wq = (g.w + VEC - 1) / VEC;
    const long total = (long)g.n * g.h * wq;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int xq = (int)(id % wq);
    const long row = id / wq;
    const int y = (int)(row % g.h), img = (int)(row / g.h);
    const Blend by = blend_of(y, g.inv_th, g.tiles_y);
    const unsigned char* limg = lut + (long)img * g.tiles_x * g.tiles_y * 256;
    const unsigned char* l1 = limg + (long)by.i1 * g.tiles_x * 256;
    const unsigned char* l2 = limg + (long)by.i2 * g.tiles_x * 256;
    const long plane = (long)g.h * g.w;
    const long pix = (long)y * g.w + (long)xq * VEC;
    const float* px = x + (long)img * 3 * plane + pix;
    float* po = out + (long)img * 3 * plane + pix;
    float c[3][VEC];
    unsigned char lv[VEC];
    if (VEC == 4) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) *(f32x4*)c[ch] = *(const f32x4*)(px + ch * plane);
        *(uchar4*)lv = *(const uchar4*)(lplane + (long)img * plane + pix);
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) c[ch][0] = px[ch * plane];
        lv[0] = lplane[(long)img * plane + pix];
    }
    const float fth = 7.787f * 0.008856f + 16.0f / 116.0f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float lin[3], f[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) lin[ch] = srgb_to_linear(add_rn(mul_rn(c[ch][k], p.in_scale[ch]), p.in_shift[ch]));
#pragma unroll
        for (int i = 0; i < 3; ++i)
            f[i] = lab_f(add_rn(add_rn(mul_rn(p.fwd[3 * i], lin[0]), mul_rn(p.fwd[3 * i + 1], lin[1])), mul_rn(p.fwd[3 * i + 2], lin[2])));
        // the reference keeps (a + 128) / 255 and (b + 128) / 255 between the two conversions (functional.py:36,60)
        const float a = sub_rn(mul_rn(div_rn(add_rn(mul_rn(500.0f, sub_rn(f[0], f[1])), 128.0f), 255.0f), 255.0f), 128.0f);
        const float b = sub_rn(mul_rn(div_rn(add_rn(mul_rn(200.0f, sub_rn(f[1], f[2])), 128.0f), 255.0f), 255.0f), 128.0f);
        const unsigned nv = blend_lut(l1, l2, blend_of(xq * VEC + k, g.inv_tw, g.tiles_x), by, lv[k]);
        const float L = mul_rn(div_rn((float)nv, 255.0f), 100.0f);
        float yv, fy;
        if (L <= 0.008856f * 903.3f) { yv = L / 903.3f; fy = 7.787f * yv + 16.0f / 116.0f; }
        else { fy = (L + 16.0f) / 116.0f; yv = fy * fy * fy; }
        const float fx = a / 500.0f + fy, fz = fy - b / 200.0f;
        const float xv = fx <= fth ? (fx - 16.0f / 116.0f) / 7.787f : fx * fx * fx;
        const float zv = fz <= fth ? (fz - 16.0f / 116.0f) / 7.787f : fz * fz * fz;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float rgb = linear_to_srgb(p.inv[3 * ch] * xv + p.inv[3 * ch + 1] * yv + p.inv[3 * ch + 2] * zv);
            c[ch][k] = div_rn(sub_rn(rgb, p.out_mean[ch]), p.out_std[ch]);
        }
    }
    if (VEC == 4) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) *(f32x4*)(po + ch * plane) = *(f32x4*)c[ch];
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) po[ch * plane] = c[ch][0];
    }
}

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

struct Layout { size_t hist, lut, lplane, total; };

int plan(int n, int h, int w, int tiles_x, int tiles_y, double clip_limit, ClaheGeom& g, Layout& L) {
    GDT_REQUIRE(n >= 1 && h >= 1 && w >= 1, "clahe needs a non-empty batch of planes");
    GDT_REQUIRE(tiles_x >= 1 && tiles_y >= 1 && tiles_x <= w && tiles_y <= h, "clahe tile grid must fit the image");
    GDT_REQUIRE((long)n * h * w < (1l << 40), "clahe batch too large");
    g.n = n; g.h = h; g.w = w; g.tiles_x = tiles_x; g.tiles_y = tiles_y;
    int eh = h, ew = w;
    if (w % tiles_x != 0 || h % tiles_y != 0) {       // cv::CLAHE::apply extends BOTH dimensions as soon as one does not divide
        eh = h + tiles_y - h % tiles_y;
        ew = w + tiles_x - w % tiles_x;
        GDT_REQUIRE(eh - h < h && ew - w < w && h >= 2 && w >= 2, "clahe: REFLECT_101 extension larger than the image");
    }
    g.th = eh / tiles_y; g.tw = ew / tiles_x;
    const long area = (long)g.th * g.tw;
    GDT_REQUIRE(area < (1l << 24), "clahe tile too large");
    g.lut_scale = 255.0f / (float)area;
    g.limit = 0;
    if (clip_limit > 0.0) {
        g.limit = (int)(clip_limit * (double)area / 256.0);
        if (g.limit < 1) g.limit = 1;
    }
    g.inv_tw = 1.0f / (float)g.tw;
    g.inv_th = 1.0f / (float)g.th;
    const int rows = (int)((4096 + g.tw - 1) / g.tw);                    // ~4096 pixels (16 per lane) per histogram workgroup
    g.rows_per_part = rows < 1 ? 1 : (rows > g.th ? g.th : rows);
    g.parts = (g.th + g.rows_per_part - 1) / g.rows_per_part;
    const size_t tiles = (size_t)n * tiles_x * tiles_y;
    size_t off = 0;
    L.hist = off; off += align_up(tiles * 256 * sizeof(unsigned));
    L.lut = off; off += align_up(tiles * 256);
    L.lplane = off; off += align_up((size_t)n * h * w);
    L.total = off + ALIGN;
    return GDT_OK;
}

inline char* aligned_base(void* ws) { return (char*)(((uintptr_t)ws + ALIGN - 1) / ALIGN * ALIGN); }

void fill_matrices(LabParams& p) {
    // OpenCV color_lab.cpp: sRGB D65 primaries and white point
    static const double rgb2xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    static const double xyz2rgb[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
    static const double white[3] = {0.950456, 1.0, 1.088754};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            p.fwd[3 * i + j] = (float)(rgb2xyz[3 * i + j] / white[i]);
            p.inv[3 * i + j] = (float)(xyz2rgb[3 * i + j] * white[j]);
        }
}

}  // namespace

extern "C" {

int gdt_clahe_workspace_bytes(int n, int h, int w, int tiles_x, int tiles_y, size_t* bytes) {
    GDT_REQUIRE(bytes != nullptr, "bytes");
    ClaheGeom g;
    Layout L;
    int rc = plan(n, h, w, tiles_x, tiles_y, 1.0, g, L);
    if (rc != GDT_OK) return rc;
    *bytes = L.total;
    return GDT_OK;
}

int gdt_clahe_u8(const unsigned char* src, unsigned char* dst, int n, int h, int w, double clip_limit, int tiles_x, int tiles_y,
                 void* workspace, size_t workspace_bytes, void* stream_) {
    ClaheGeom g;
    Layout L;
    int rc = plan(n, h, w, tiles_x, tiles_y, clip_limit, g, L);
    if (rc != GDT_OK) return rc;
    GDT_REQUIRE(src != nullptr && dst != nullptr && workspace != nullptr, "clahe: null buffer");
    if (workspace_bytes < L.total) { gdt_set_error("clahe: workspace too small"); return GDT_ERR_WORKSPACE; }
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = aligned_base(workspace);
    unsigned* hist = (unsigned*)(ws + L.hist);
    unsigned char* lut = (unsigned char*)(ws + L.lut);
    const int tiles = n * tiles_x * tiles_y;
    GDT_CHECK_HIP(hipMemsetAsync(hist, 0, (size_t)tiles * 256 * sizeof(unsigned), stream));
    LabParams p = {};
    hipLaunchKernelGGL(clahe_hist_kernel<false>, dim3(tiles * g.parts), dim3(256), 0, stream, g, src, (const float*)nullptr, p,
                       (unsigned char*)nullptr, hist);
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles), dim3(256), 0, stream, g, hist, lut);
    const bool vec = w % 4 == 0 && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0);
    const long work = (long)n * h * (vec ? w / 4 : w);
    if (vec) hipLaunchKernelGGL(clahe_apply_u8_kernel<4>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, g, src, lut, dst);
    else hipLaunchKernelGGL(clahe_apply_u8_kernel<1>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, g, src, lut, dst);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_clahe_lab_f32(const float* x, float* y, int n, int h, int w, const float* in_scale, const float* in_shift, const float* out_mean,
                      const float* out_std, double clip_limit, int tiles_x, int tiles_y, void* workspace, size_t workspace_bytes,
                      void* stream_) {
    ClaheGeom g;
    Layout L;
    int rc = plan(n, h, w, tiles_x, tiles_y, clip_limit, g, L);
    if (rc != GDT_OK) return rc;
    GDT_REQUIRE(x != nullptr && y != nullptr && workspace != nullptr, "clahe: null buffer");
    if (workspace_bytes < L.total) { gdt_set_error("clahe: workspace too small"); return GDT_ERR_WORKSPACE; }
    LabParams p;
    for (int c = 0; c < 3; ++c) {
        p.in_scale[c] = in_scale ? in_scale[c] : 1.f;
        p.in_shift[c] = in_shift ? in_shift[c] : 0.f;
        p.out_mean[c] = out_mean ? out_mean[c] : 0.f;
        p.out_std[c] = out_std ? out_std[c] : 1.f;
        GDT_REQUIRE(p.out_std[c] != 0.f, "clahe: zero output std");
    }
    fill_matrices(p);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = aligned_base(workspace);
    unsigned* hist = (unsigned*)(ws + L.hist);
    unsigned char* lut = (unsigned char*)(ws + L.lut);
    unsigned char* lplane = (unsigned char*)(ws + L.lplane);
    const int tiles = n * tiles_x * tiles_y;
    GDT_CHECK_HIP(hipMemsetAsync(hist, 0, (size_t)tiles * 256 * sizeof(unsigned), stream));
    hipLaunchKernelGGL(clahe_hist_kernel<true>, dim3(tiles * g.parts), dim3(256), 0, stream, g, (const unsigned char*)nullptr, x, p, lplane, hist);
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles), dim3(256), 0, stream, g, hist, lut);
    const bool vec = w % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    const long work = (long)n * h * (vec ? w / 4 : w);
    if (vec) hipLaunchKernelGGL(clahe_apply_lab_kernel<4>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, g, x, lplane, lut, p, y);
    else hipLaunchKernelGGL(clahe_apply_lab_kernel<1>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, g, x, lplane, lut, p, y);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // extern "C"
