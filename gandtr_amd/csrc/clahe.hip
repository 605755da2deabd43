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
// default -- so the exact steps use the local helpers below, compiled with contraction off.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }

#pragma clang fp contract(fast)      // everything that is not written with the helpers above may fuse

struct ClaheGeom {
    int n, h, w, tiles_x, tiles_y, th, tw;      // tile size in pixels of the extended image
    int limit;                                  // integer clip limit per bin (0: no clipping)
    float lut_scale, inv_tw, inv_th;
    int rows_per_part, parts;                   // histogram pass: tile rows per workgroup, workgroups per tile
};

struct LabParams {
    float in_scale[3], in_shift[3];             // rgb = x * in_scale + in_shift
    float out_mean[3], out_std[3];              // y = (rgb - out_mean) * out_std   (out_std = 1 / std)
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

// Clip, redistribute, scan, scale (CLAHE_CalcLut_Body of OpenCV's clahe.cpp) by ONE wave: lane l owns bins 4l .. 4l+3 and returns
// their four table entries packed little-endian.  Integer results throughout; the two divisions are float reciprocals whose
// quotients stay >= 0.5 / 256 away from an integer boundary (or are exact powers of two), so truncation gives the exact integer.
__device__ __forceinline__ unsigned lut_quad(int h0, int h1, int h2, int h3, const ClaheGeom& g) {
    const int lane = threadIdx.x & 63;
    int hv[4] = {h0, h1, h2, h3};
    if (g.limit > 0) {
        int excess = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { excess += max(hv[j] - g.limit, 0); hv[j] = min(hv[j], g.limit); }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) excess += __shfl_xor(excess, m);
        const int batch = excess >> 8, residual = excess & 255;
        int step = 1;
        float rstep = 1.f;
        if (residual) {
            step = max((int)(256.0f * __builtin_amdgcn_rcpf((float)residual) + 1e-3f), 1);      // 256 / residual
            rstep = __builtin_amdgcn_rcpf((float)step);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bin = 4 * lane + j;
            const int k = (int)(((float)bin + 0.5f) * rstep);                                     // bin / step
            hv[j] += batch + ((residual && k * step == bin && k < residual) ? 1 : 0);
        }
    }
    hv[1] += hv[0]; hv[2] += hv[1]; hv[3] += hv[2];                                               // inclusive within the lane
    int run = hv[3];                                                                              // inclusive scan of lane totals
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int t = __shfl_up(run, m);
        if (lane >= m) run += t;
    }
    const int base = run - hv[3];
    unsigned packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float f = __builtin_rintf(mul_rn((float)(base + hv[j]), g.lut_scale));
        packed |= (unsigned)fminf(fmaxf(f, 0.f), 255.f) << (8 * j);
    }
    return packed;
}

// FUSED (one workgroup owns the whole tile): the lookup table is finished here and no global histogram exists.
// QUAD (RGB only; w, tile width multiples of 4, no border extension): four pixels of a row per lane through 16-byte loads.
template <bool RGB, bool FUSED, bool QUAD>
__global__ __launch_bounds__(256) void clahe_hist_kernel(ClaheGeom g, const unsigned char* __restrict__ src8, const float* __restrict__ x,
                                                         LabParams p, unsigned char* __restrict__ lplane, unsigned* __restrict__ hist,
                                                         unsigned char* __restrict__ lut) {
    __shared__ __attribute__((aligned(16))) unsigned sh[4][256];
    const int tid = threadIdx.x, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k][tid] = 0;
    __syncthreads();
    const int tiles = g.tiles_x * g.tiles_y;
    const int part = blockIdx.x % g.parts, tile = (blockIdx.x / g.parts) % tiles, img = blockIdx.x / (g.parts * tiles);
    const int ty = tile / g.tiles_x, tx = tile % g.tiles_x;
    const int r0 = part * g.rows_per_part, r1 = min(r0 + g.rows_per_part, g.th);
    const unsigned plane = (unsigned)g.h * g.w;
    const float* px = x + (size_t)img * 3 * plane;
    const unsigned char* p8 = src8 + (size_t)img * plane;
    unsigned char* pl = lplane + (size_t)img * plane;
    if (QUAD) {
        const int tq = g.tw >> 2, count = (r1 - r0) * tq;
        const float inv_tq = 4.0f * g.inv_tw;
        for (int i = tid; i < count; i += 256) {
            const int rr = (int)(((float)i + 0.5f) * inv_tq);                         // i / tq (see lut_quad for the argument)
            const unsigned off = (unsigned)(ty * g.th + r0 + rr) * g.w + tx * g.tw + 4 * (i - rr * tq);
            f32x4 c[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) c[ch] = *(const f32x4*)(px + (off + ch * plane));
            unsigned packed = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned v = lightness_u8(c[0][k], c[1][k], c[2][k], p);
                packed |= v << (8 * k);
                atomicAdd(&sh[wave][v], 1u);
            }
            *(unsigned*)(pl + off) = packed;
        }
    } else {
        const int count = (r1 - r0) * g.tw;
        for (int i0 = 0; i0 < count; i0 += 1024) {
            // four pixels per lane with their loads issued together (masked lanes re-read pixel 0 of the tile part and drop it)
            unsigned off[4];
            bool ok[4], inside[4];
            float r[4], gch[4], b[4];
            unsigned v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * 256 + tid;
                ok[j] = i < count;
                const int ii = ok[j] ? i : 0;
                const int rr = (int)(((float)ii + 0.5f) * g.inv_tw), c = ii - rr * g.tw;     // ii / tw
                const int ey = ty * g.th + r0 + rr, ex = tx * g.tw + c;
                const int sy = ey < g.h ? ey : 2 * (g.h - 1) - ey, sx = ex < g.w ? ex : 2 * (g.w - 1) - ex;      // BORDER_REFLECT_101
                inside[j] = ok[j] && ey < g.h && ex < g.w;
                off[j] = (unsigned)sy * g.w + sx;
                if (RGB) { r[j] = px[off[j]]; gch[j] = px[off[j] + plane]; b[j] = px[off[j] + 2 * plane]; }
                else v[j] = p8[off[j]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (RGB) {
                    v[j] = lightness_u8(r[j], gch[j], b[j], p);
                    if (inside[j]) pl[off[j]] = (unsigned char)v[j];
                }
                if (ok[j]) atomicAdd(&sh[wave][v[j]], 1u);
            }
        }
    }
    __syncthreads();
    if (FUSED) {
        if (wave != 0) return;
        const uint4 a0 = *(const uint4*)&sh[0][4 * tid], a1 = *(const uint4*)&sh[1][4 * tid];
        const uint4 a2 = *(const uint4*)&sh[2][4 * tid], a3 = *(const uint4*)&sh[3][4 * tid];
        *(unsigned*)(lut + ((size_t)img * tiles + tile) * 256 + 4 * tid) =
            lut_quad((int)(a0.x + a1.x + a2.x + a3.x), (int)(a0.y + a1.y + a2.y + a3.y), (int)(a0.z + a1.z + a2.z + a3.z),
                     (int)(a0.w + a1.w + a2.w + a3.w), g);
    } else {
        const unsigned s = sh[0][tid] + sh[1][tid] + sh[2][tid] + sh[3][tid];
        if (s) atomicAdd(&hist[((size_t)img * tiles + tile) * 256 + tid], s);
    }
}

// row-split tiles (few, large tiles): one wave per (image, tile) turns the merged global histogram into the lookup table
__global__ __launch_bounds__(64) void clahe_lut_kernel(ClaheGeom g, const unsigned* __restrict__ hist, unsigned char* __restrict__ lut) {
    const uint4 hv = *(const uint4*)(hist + (size_t)blockIdx.x * 256 + 4 * threadIdx.x);
    *(unsigned*)(lut + (size_t)blockIdx.x * 256 + 4 * threadIdx.x) = lut_quad((int)hv.x, (int)hv.y, (int)hv.z, (int)hv.w, g);
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

// CLAHE_Interpolation_Body: unfused float32, round half to even, saturate.  `limg`: the image's tables (workgroup-uniform base,
// 32-bit lane offsets); row1 / row2 = by.i1 / by.i2 * tiles_x.
__device__ __forceinline__ unsigned blend_lut(const unsigned char* __restrict__ limg, unsigned row1, unsigned row2, const Blend& bx,
                                              const Blend& by, unsigned v) {
    const float v11 = (float)limg[(row1 + bx.i1) * 256u + v], v12 = (float)limg[(row1 + bx.i2) * 256u + v];
    const float v21 = (float)limg[(row2 + bx.i1) * 256u + v], v22 = (float)limg[(row2 + bx.i2) * 256u + v];
    const float top = add_rn(mul_rn(v11, bx.a1), mul_rn(v12, bx.a));
    const float bot = add_rn(mul_rn(v21, bx.a1), mul_rn(v22, bx.a));
    const float res = __builtin_rintf(add_rn(mul_rn(top, by.a1), mul_rn(bot, by.a)));
    return (unsigned)fminf(fmaxf(res, 0.f), 255.f);
}

// 8-bit plane in, 8-bit plane out; VEC pixels of one row per lane; blockIdx.y = image
template <int VEC>
__global__ __launch_bounds__(256) void clahe_apply_u8_kernel(ClaheGeom g, const unsigned char* __restrict__ src, const unsigned char* __restrict__ lut,
                                                             unsigned char* __restrict__ dst) {
    const unsigned wq = (unsigned)(g.w + VEC - 1) / VEC;
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= wq * (unsigned)g.h) return;
    const unsigned y = q / wq, xq = q - y * wq;
    const Blend by = blend_of((int)y, g.inv_th, g.tiles_y);
    const unsigned char* limg = lut + (size_t)blockIdx.y * g.tiles_x * g.tiles_y * 256;
    const unsigned row1 = by.i1 * g.tiles_x, row2 = by.i2 * g.tiles_x;
    const size_t plane = (size_t)g.h * g.w;
    const unsigned pix = y * (unsigned)g.w + xq * VEC;
    const unsigned char* ps = src + blockIdx.y * plane;
    unsigned char* pd = dst + blockIdx.y * plane;
    unsigned char v[VEC];
    if (VEC == 4) *(uchar4*)v = *(const uchar4*)(ps + pix);
    else v[0] = ps[pix];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = (unsigned char)blend_lut(limg, row1, row2, blend_of((int)(xq * VEC + k), g.inv_tw, g.tiles_x), by, v[k]);
    if (VEC == 4) *(uchar4*)(pd + pix) = *(uchar4*)v;
    else pd[pix] = v[0];
}

// fp32 NCHW RGB in / out with the lightness plane of pass 1; blockIdx.y = image.  Nothing here has to be bit-exact (the 8-bit
// lightness was fixed by pass 1, the LUT blend goes through the exact helpers), so the colour arithmetic may fuse.
template <int VEC>
__global__ __launch_bounds__(256) void clahe_apply_lab_kernel(ClaheGeom g, const float* __restrict__ x, const unsigned char* __restrict__ lplane,
                                                              const unsigned char* __restrict__ lut, LabParams p, float* __restrict__ out) {
    const unsigned wq = (unsigned)(g.w + VEC - 1) / VEC;
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= wq * (unsigned)g.h) return;
    const unsigned y = q / wq, xq = q - y * wq;
    const Blend by = blend_of((int)y, g.inv_th, g.tiles_y);
    const unsigned char* limg = lut + (size_t)blockIdx.y * g.tiles_x * g.tiles_y * 256;
    const unsigned row1 = by.i1 * g.tiles_x, row2 = by.i2 * g.tiles_x;
    const unsigned plane = (unsigned)g.h * g.w;                     // 3 * plane * 4 bytes < 2^32 (checked by the launcher)
    const unsigned pix = y * (unsigned)g.w + xq * VEC;
    const float* px = x + (size_t)blockIdx.y * 3 * plane;
    float* po = out + (size_t)blockIdx.y * 3 * plane;
    const unsigned char* pl = lplane + (size_t)blockIdx.y * plane;
    float c[3][VEC];
    unsigned char lv[VEC];
    if (VEC == 4) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) *(f32x4*)c[ch] = *(const f32x4*)(px + (pix + ch * plane));
        *(uchar4*)lv = *(const uchar4*)(pl + pix);
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) c[ch][0] = px[pix + ch * plane];
        lv[0] = pl[pix];
    }
    const float fth = 7.787f * 0.008856f + 16.0f / 116.0f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        float lin[3], f[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) lin[ch] = srgb_to_linear(c[ch][k] * p.in_scale[ch] + p.in_shift[ch]);
#pragma unroll
        for (int i = 0; i < 3; ++i) f[i] = lab_f(p.fwd[3 * i] * lin[0] + p.fwd[3 * i + 1] * lin[1] + p.fwd[3 * i + 2] * lin[2]);
        // (the reference's (a + 128) / 255 * 255 - 128 round trip between the conversions, functional.py:36,60, is the identity up to
        // 1e-5 in a / b and is not replayed; constant divisions are reciprocal multiplies -- this half is compared with a tolerance)
        const float a500 = f[0] - f[1], b200 = f[1] - f[2];                      // a / 500, b / 200
        const unsigned nv = blend_lut(limg, row1, row2, blend_of((int)(xq * VEC + k), g.inv_tw, g.tiles_x), by, lv[k]);
        const float L = (float)nv * (100.0f / 255.0f);
        float yv, fy;
        if (L <= 0.008856f * 903.3f) { yv = L * (1.0f / 903.3f); fy = 7.787f * yv + 16.0f / 116.0f; }
        else { fy = (L + 16.0f) * (1.0f / 116.0f); yv = fy * fy * fy; }
        const float fx = a500 + fy, fz = fy - b200;
        const float xv = fx <= fth ? (fx - 16.0f / 116.0f) * (1.0f / 7.787f) : fx * fx * fx;
        const float zv = fz <= fth ? (fz - 16.0f / 116.0f) * (1.0f / 7.787f) : fz * fz * fz;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float rgb = linear_to_srgb(p.inv[3 * ch] * xv + p.inv[3 * ch + 1] * yv + p.inv[3 * ch + 2] * zv);
            c[ch][k] = (rgb - p.out_mean[ch]) * p.out_std[ch];       // out_std holds 1 / std
        }
    }
    if (VEC == 4) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) *(f32x4*)(po + (pix + ch * plane)) = *(f32x4*)c[ch];
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) po[pix + ch * plane] = c[ch][0];
    }
}

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

struct Layout { size_t hist, lut, lplane, total; };

int plan(int n, int h, int w, int tiles_x, int tiles_y, double clip_limit, ClaheGeom& g, Layout& L) {
    GDT_REQUIRE(n >= 1 && h >= 1 && w >= 1, "clahe needs a non-empty batch of planes");
    GDT_REQUIRE(tiles_x >= 1 && tiles_y >= 1 && tiles_x <= w && tiles_y <= h, "clahe tile grid must fit the image");
    GDT_REQUIRE((long)h * w < (1l << 28) && n <= 65535, "clahe: at most 2^28 pixels per image and 65535 images per call");
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
    // histogram pass: one workgroup per tile (lookup table finished in the same launch) when that fills the chip or the tiles are
    // small; otherwise ~4096 pixels (16 per lane) per workgroup, merged through global atomics
    const int rows = (int)((4096 + g.tw - 1) / g.tw);
    g.rows_per_part = rows < 1 ? 1 : (rows > g.th ? g.th : rows);
    if ((long)n * tiles_x * tiles_y >= 1024 || area <= 4096) g.rows_per_part = g.th;
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
    LabParams p = {};
    if (g.parts == 1) {
        hipLaunchKernelGGL((clahe_hist_kernel<false, true, false>), dim3(tiles), dim3(256), 0, stream, g, src, (const float*)nullptr, p,
                           (unsigned char*)nullptr, hist, lut);
    } else {
        GDT_CHECK_HIP(hipMemsetAsync(hist, 0, (size_t)tiles * 256 * sizeof(unsigned), stream));
        hipLaunchKernelGGL((clahe_hist_kernel<false, false, false>), dim3(tiles * g.parts), dim3(256), 0, stream, g, src, (const float*)nullptr, p,
                           (unsigned char*)nullptr, hist, lut);
        hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles), dim3(64), 0, stream, g, hist, lut);
    }
    const bool vec = w % 4 == 0 && ((uintptr_t)src % 4 == 0) && ((uintptr_t)dst % 4 == 0);
    const long work = (long)h * (vec ? w / 4 : w);                       // lanes per image; blockIdx.y = image
    const dim3 grid((unsigned)((work + 255) / 256), (unsigned)n);
    if (vec) hipLaunchKernelGGL(clahe_apply_u8_kernel<4>, grid, dim3(256), 0, stream, g, src, lut, dst);
    else hipLaunchKernelGGL(clahe_apply_u8_kernel<1>, grid, dim3(256), 0, stream, g, src, lut, dst);
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
        GDT_REQUIRE(!out_std || out_std[c] != 0.f, "clahe: zero output std");
        p.out_std[c] = out_std ? 1.f / out_std[c] : 1.f;
    }
    fill_matrices(p);
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = aligned_base(workspace);
    unsigned* hist = (unsigned*)(ws + L.hist);
    unsigned char* lut = (unsigned char*)(ws + L.lut);
    unsigned char* lplane = (unsigned char*)(ws + L.lplane);
    const int tiles = n * tiles_x * tiles_y;
    const bool vec = w % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    const bool quad = vec && g.tw % 4 == 0 && g.tw * tiles_x == w && g.th * tiles_y == h;
    const unsigned char* no8 = nullptr;
    if (g.parts == 1) {
        if (quad) hipLaunchKernelGGL((clahe_hist_kernel<true, true, true>), dim3(tiles), dim3(256), 0, stream, g, no8, x, p, lplane, hist, lut);
        else hipLaunchKernelGGL((clahe_hist_kernel<true, true, false>), dim3(tiles), dim3(256), 0, stream, g, no8, x, p, lplane, hist, lut);
    } else {
        GDT_CHECK_HIP(hipMemsetAsync(hist, 0, (size_t)tiles * 256 * sizeof(unsigned), stream));
        if (quad) hipLaunchKernelGGL((clahe_hist_kernel<true, false, true>), dim3(tiles * g.parts), dim3(256), 0, stream, g, no8, x, p, lplane, hist, lut);
        else hipLaunchKernelGGL((clahe_hist_kernel<true, false, false>), dim3(tiles * g.parts), dim3(256), 0, stream, g, no8, x, p, lplane, hist, lut);
        hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles), dim3(64), 0, stream, g, hist, lut);
    }
    const long work = (long)h * (vec ? w / 4 : w);                       // lanes per image; blockIdx.y = image
    const dim3 grid((unsigned)((work + 255) / 256), (unsigned)n);
    if (vec) hipLaunchKernelGGL(clahe_apply_lab_kernel<4>, grid, dim3(256), 0, stream, g, x, lplane, lut, p, y);
    else hipLaunchKernelGGL(clahe_apply_lab_kernel<1>, grid, dim3(256), 0, stream, g, x, lplane, lut, p, y);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // extern "C"
