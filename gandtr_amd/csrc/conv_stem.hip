// First layer of every network on the path: KxK convolution from an image (3 channels, padded to 8) to 64 channels
//   generator stem  ReflectionPad2d(3) + Conv2d(3, 64, 7) (+ InstanceNorm statistics)         p2p_networks.py:269-272
//   ResNet-101 stem Conv2d(3, 64, 7, stride 2, pad 3) + BN + ReLU                             torchvision resnet.py, sliced at
//   VGG16 conv1_1   Conv2d(3, 64, 3, pad 1) + ReLU                                            imageretrievalnet.py:185-190
// (gfx950, MI355X).  The generic implicit GEMM gathers this layer's A operand 16 bytes at a time (every 8-channel piece is
// its own tap) and re-reads each input pixel K*K times from L2: 0.50 ms for the 79 GFLOP generator stem, 0.97 ms for the
// ResNet stem, against ~0.12 ms of output traffic.  Here a pixel is ONE 16-byte LDS word:
//   * a workgroup owns TH x 32 output pixels (TH = 4 rows per wave x 4 waves at stride 1, 2 x 4 at stride 2); the input halo
//     ((TH-1)*S + K) x (31*S + K) pixels, 13-23 KB) is staged once by LDS-DMA, padding resolved in the source address;
//   * K index = tap * 8 + channel, so one MFMA k-step (16) is two taps: the A fragment of lane (pixel fr, half fh) is the
//     16-byte word of pixel (oy*S + ty, ox*S + tx) with tap = 2*ks + fh -- a plain ds_read_b128 at a per-lane base plus a
//     compile-time offset (two bases: the second tap of a pair is either the next pixel or the start of the next halo row);
//   * the weights (64 x K*K*8, up to 50 KB) sit in LDS in B-fragment order for the lifetime of the persistent workgroup;
//   * MFMA operands swapped (D = W * A^T) so a lane holds 4 consecutive output channels: the epilogue transposes 32-pixel
//     blocks through a wave-private LDS patch with 8-byte writes and stores whole 4 KB row segments; InstanceNorm
//     statistics (one 128-pixel record per wave) come from the stored fp16 values, reduced in a fixed order.
// Two workgroups share a CU (one loads while the other computes).
#include <cstdio>
#include <cstdlib>

#include <type_traits>

#include "gdt_common.h"

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

#ifndef GDT_STEM_PF_AUG
#define GDT_STEM_PF_AUG 3
#endif
constexpr int TW = 32;                     // output tile width (one 32-pixel MFMA block per output row)
constexpr int NWAVE = 4, NT = NWAVE * 64;
constexpr int CP = 72;                     // halves per pixel row of the wave-private transpose patch (64 + 8)
constexpr int PATCH_BYTES = 32 * CP * 2;   // 4608 B per wave

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_dst, 16, 0, 0);
}

template <int KS, int S>
struct StemCfg {
    static constexpr int RPW = S == 1 ? 4 : 2;                   // output rows per wave
    static constexpr int TH = RPW * NWAVE;                       // output rows per tile
    static constexpr int HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
    static constexpr int NTAPS = KS * KS, NKS = (NTAPS + 1) / 2;
    static constexpr int HPIX = HH * HW;
    static constexpr int HROWS_ALLOC = HH + 1;                   // one spare (zeroed) row: the odd tap past the end reads it
    static constexpr int HB0 = HROWS_ALLOC * HW * 16, HB1 = (HPIX + 63) / 64 * 64 * 16;       // (the last DMA round is a whole 64 pixels)
    static constexpr int HBYTES = (((HB0 > HB1 ? HB0 : HB1) + 1023) / 1024) * 1024;
    static constexpr int WBYTES = NKS * 2 * 1024;                // B fragments: [ks][j][lane][8 halves]
    static constexpr int EPI_BYTES = NWAVE * PATCH_BYTES;
    static constexpr int LDS = WBYTES + (HBYTES > EPI_BYTES ? HBYTES : EPI_BYTES);
};

// AUG ("f16c" precision mode, fp32-class result on the same two-taps-per-k-step machinery): the 8 slots of a pixel word carry
// [a_hi (cin values), (a - a_hi) * 2^8 (cin values), 0 ..] (written by the input pack kernel), the weight slots of a tap
// [w_hi, w_hi * 2^-8, 0 ..] -- so the activation's fp16 rounding residual rides in the channel padding of the SAME MFMA -- and a second
// MFMA per fragment pair with W2 = [w - w_hi, 0 ..] (fp16 subnormals keep 7-8 bits of it: plenty for a 2^-12 correction) adds the
// weight residual.  W2 streams from L2 into registers.  Output fp32 NHWC, statistics from the fp32 values.
template <int KS, int S, bool AUG = false>
__global__ __launch_bounds__(NT, 2) void conv_stem_kernel(const ConvLaunch d, const int ntiles) {
    using C = StemCfg<KS, S>;
    constexpr int RPW = C::RPW, TH = C::TH, HW = C::HW, NKS = C::NKS, PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wlds = smem;                         // weights
    char* hbuf = smem + C::WBYTES;             // halo, later the transpose patches
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int tiles_x = (d.OW + TW - 1) / TW, tiles_y = (d.OH + TH - 1) / TH, tpi = tiles_x * tiles_y;

    // XCD-chunked persistent schedule (as conv_head7.hip)
    const int per_xcd = (ntiles + 7) >> 3, SS = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int span_lo = xcd * per_xcd, span_hi = min(span_lo + per_xcd, ntiles);
    int tile = span_lo + slot;
    if (tile >= span_hi) return;

    // weights -> LDS (fragment order, linear copy), spare halo row zeroed
    for (int i = tid; i < C::WBYTES / 16; i += NT) *(float4*)(wlds + i * 16) = *(const float4*)((const char*)d.w_frag + i * 16);
    for (int i = tid; i < C::HBYTES / 16; i += NT) *(float4*)(hbuf + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool refl = d.pad_reflect != 0;

    // A fragment: pixel (row block i of this wave = output row wave*RPW + i, column fr), tap 2*ks + fh
    const int a_base = ((wave * RPW * S) * HW + fr * S) * 16;
    const int a_b16 = a_base + fh * 16;                          // second tap of the pair is the next pixel ...
    const int a_bwrap = a_base + fh * (HW - (KS - 1)) * 16;      // ... or the first pixel of the next halo row

    for (;;) {
        const int n = tile / tpi, r = tile - n * tpi;
        const int y0 = (r / tiles_x) * TH, x0 = (r % tiles_x) * TW;
        __syncthreads();                               // previous tile's transpose patches are done with the buffer
        // ---- halo: one 16-byte word per pixel, 64 pixels per DMA instruction
#pragma unroll 1
        for (int j = 0; j < (C::HPIX + NT - 1) / NT; ++j) {
            const int hp0 = min((j * NWAVE + wave) * 64, (C::HPIX - 1) / 64 * 64);      // (surplus rounds repeat the last one)
            const int hp = min(hp0 + lane, C::HPIX - 1);
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = y0 * S - PAD + hy, ix = x0 * S - PAD + hx;
            int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
            int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
            ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);
            const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
            const f16* src = d.in + (long)((n * d.H + ry) * d.W + rx) * 8;
            glds16((inb | refl) ? src : d.zeros, hbuf + hp0 * 16);
        }
        __syncthreads();                               // DMA landed (the barrier drains vmcnt) and visible

        f32x16 acc[RPW][2];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        {
            constexpr int PF = AUG ? GDT_STEM_PF_AUG : 3;      // fragment sets in flight (the f16c form carries a second weight set: 256 registers)
            f16x8 af[PF][RPW], bf[PF][2], b2[PF][2];
            auto frags = [&](int ks, f16x8 (&a)[RPW], f16x8 (&b)[2], f16x8 (&bb)[2]) {
                const int t0 = 2 * ks, ty = t0 / KS, tx = t0 - ty * KS;
                const bool wrap = tx == KS - 1;                // tap t0 + 1 starts the next kernel row
                const int base = wrap ? a_bwrap : a_b16;
#pragma unroll
                for (int i = 0; i < RPW; ++i) a[i] = *(const f16x8*)(hbuf + base + ((i * S + ty) * HW + tx) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *(const f16x8*)(wlds + ((ks * 2 + j) * 64 + lane) * 16);
                if (AUG) {          // uniform base + 32-bit lane offset (opaque: keeps the zero-extension next to the load, i.e. the
                    unsigned lo = lane * 16;        //  scalar-base addressing form; per-lane 64-bit addresses for 50 fragments spilled 34 registers)
                    asm volatile("" : "+v"(lo));
#pragma unroll
                    for (int j = 0; j < 2; ++j) bb[j] = *(const f16x8*)((const char*)d.w_frag2 + (size_t)((ks * 2 + j) * 1024) + lo);
                }
            };
#pragma unroll
            for (int p = 0; p < PF - 1; ++p) frags(p, af[p], bf[p], b2[p]);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + PF - 1 < NKS) frags(ks + PF - 1, af[(ks + PF - 1) % PF], bf[(ks + PF - 1) % PF], b2[(ks + PF - 1) % PF]);
#pragma unroll
                for (int i = 0; i < RPW; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                    {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks % PF][j], af[ks % PF][i], acc[i][j], 0, 0, 0);   // D[cout][pixel]
                        if (AUG) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2[ks % PF][j], af[ks % PF][i], acc[i][j], 0, 0, 0);
                    }
                if (AUG) __builtin_amdgcn_sched_barrier(0);        // (keeps the residual-weight loads of later steps from being hoisted: 70 spilled registers)
            }
        }
        __syncthreads();                               // the halo has been consumed by every wave

        if (AUG) {
            // ---- fp32 epilogue: every 32 x 32 block through the wave's private 4 KB patch (XOR-swizzled, conflict-free both ways), read
            // back with 8 lanes per pixel = whole 128-byte lines per store; a lane then owns 4 channels of 4 pixels per block
            float* fpatch = (float*)(hbuf + wave * PATCH_BYTES);
            float* __restrict__ outp = (float*)d.out;
            const int pl = lane >> 3, q = lane & 7, wswz = ((fr >> 1) & 7) << 2;
            // interior tiles (all of them when OH, OW are multiples of the tile) run a branch-free body: no per-store bounds test,
            // statistics always accumulated, ReLU as a max with 0 or -inf
            const bool full_tile = (y0 + TH <= d.OH) & (x0 + TW <= d.OW);
            const float lo = d.relu ? 0.f : -__builtin_inff();
            const unsigned obase = (unsigned)((n * d.OH + y0 + wave * RPW) * d.OW + x0 + pl) * 64u + 4u * q;      // (< 2^32 elements: eligibility)
            auto epilogue = [&](auto fast_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int colq = j * 32 + 4 * q;
                    const float4 bv = d.bias ? *(const float4*)(d.bias + colq) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < RPW; ++i) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x16& a = acc[i][j];
                            *(float4*)(fpatch + fr * 32 + ((8 * g + 4 * fh) ^ wswz)) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                        }
                        const int oy = y0 + wave * RPW + i;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int row = 8 * k + pl;
                            float4 v = *(const float4*)(fpatch + row * 32 + ((4 * q) ^ (((row >> 1) & 7) << 2)));
                            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                            const unsigned o = obase + (unsigned)((i * d.OW + 8 * k) * 64 + j * 32);
                            if (FAST) {
                                t1[0] += v.x; t1[1] += v.y; t1[2] += v.z; t1[3] += v.w;
                                t2[0] += v.x * v.x; t2[1] += v.y * v.y; t2[2] += v.z * v.z; t2[3] += v.w * v.w;
                                v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);
                                *(float4*)(outp + o) = v;
                            } else if ((oy < d.OH) & (x0 + row < d.OW)) {
                                if (d.stats) {
                                    t1[0] += v.x; t1[1] += v.y; t1[2] += v.z; t1[3] += v.w;
                                    t2[0] += v.x * v.x; t2[1] += v.y * v.y; t2[2] += v.z * v.z; t2[3] += v.w * v.w;
                                }
                                if (d.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                                *(float4*)(outp + o) = v;
                            }
                        }
                    }
                    if (d.stats) {      // one 128-pixel record per wave (RPW == 4), merged over the 8 pixel lanes in a fixed butterfly order
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int msk = 8; msk < 64; msk <<= 1) { t1[e] += __shfl_xor(t1[e], msk); t2[e] += __shfl_xor(t2[e], msk); }
                        if (pl == 0) {
                            float* dst = d.stats + ((long)(d.stats_tile_base + n * (tpi * NWAVE) + r * NWAVE + wave) * 2) * 64 + colq;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { dst[e] = t1[e]; dst[64 + e] = t2[e]; }
                        }
                    }
                }
            };
            if (full_tile) epilogue(std::true_type()); else epilogue(std::false_type());
            tile += SS;
            if (tile >= span_hi) break;
            continue;
        }
        // ---- epilogue: per 32-pixel row block through the wave's private patch
        f16* patch = (f16*)(hbuf + wave * PATCH_BYTES);
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = j * 32 + 8 * g + 4 * fh;
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (d.bias) bv = *(const float4*)(d.bias + col);
                    float v0 = acc[i][j][4 * g] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y, v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
                    if (d.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    f16x4 h; h[0] = (f16)v0; h[1] = (f16)v1; h[2] = (f16)v2; h[3] = (f16)v3;
                    *(f16x4*)(patch + fr * CP + col) = h;
                }
            const int oy = y0 + wave * RPW + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane + 64 * q, px = idx >> 3, ch = idx & 7;
                const f16x8 v = *(const f16x8*)(patch + px * CP + ch * 8);
                const bool ok = (oy < d.OH) & (x0 + px < d.OW);
                if (d.stats && ok) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s1[e] += f; s2[e] += f * f; }
                }
                if (ok) *(f16x8*)(d.out + ((long)((n * d.OH + oy) * d.OW + x0 + px) * 64 + ch * 8)) = v;
            }
        }
        if (d.stats) {      // one 128-pixel record per wave (RPW == 4): lanes sharing a channel group in a fixed butterfly order
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int msk = 8; msk < 64; msk <<= 1) { s1[e] += __shfl_xor(s1[e], msk); s2[e] += __shfl_xor(s2[e], msk); }
            if (lane < 8) {
                float* dst = d.stats + ((long)(d.stats_tile_base + n * (tpi * NWAVE) + r * NWAVE + wave) * 2) * 64 + lane * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { dst[e] = s1[e]; dst[64 + e] = s2[e]; }
            }
        }
        tile += SS;
        if (tile >= span_hi) break;
    }
}

template <int KS, int S, bool AUG = false>
int launch_stem(const ConvLaunch& d, hipStream_t stream) {
    using C = StemCfg<KS, S>;
    static_assert(2 * C::LDS <= 160 * 1024, "two workgroups per CU");
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_stem_kernel<KS, S, AUG>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    const int ntiles = d.N * ((d.OW + TW - 1) / TW) * ((d.OH + C::TH - 1) / C::TH);
    const int grid = min(2 * cus, (ntiles + 7) / 8 * 8);
    hipLaunchKernelGGL((conv_stem_kernel<KS, S, AUG>), dim3(grid), dim3(NT), C::LDS, stream, d, ntiles);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// ResNet stem, direct form (fp16 mode): Conv2d(3, 64, 7, stride 2, pad 3) + folded BN + ReLU straight from the caller's fp32 NCHW image --
// the input pack kernel (0.26 ms per 32 x 1024^2 batch, a 537 MB tensor written and read back) is gone -- with a denser K packing:
//   * an LDS word holds TWO horizontally adjacent pixels x 4 channel slots (3 real): word[hy][hx] = {pixel hx, pixel hx + 1}.  One MFMA k-step
//     (16) = four taps of one kernel row: lanes fh = 0 read the word at column 2 ox + 4 h (taps 4h, 4h + 1), lanes fh = 1 the word at
//     2 ox + 4 h + 2 (taps 4h + 2, 4h + 3); a kernel row is two k-steps (the eighth tap has zero weights): 14 k-steps instead of 25;
//   * the halo (21 x 70 pixels) is fetched as three coalesced fp32 loads per pixel, per-channel affine of the input op applied (the same
//     expression as the pack kernel: identical bits), rounded to fp16 and written twice (as the low half of its own word and the high half of
//     its left neighbour's); the loads of the NEXT tile are issued before the MFMAs of this one and land under them and the epilogue.
// Everything else (swapped operands, wave-private transpose patch, 128-byte line stores) is the fp16 stem above.

struct StemPairArgs { const float* x; int C; int perm[4]; float scale[4], shift[4]; };

// <7, 2>: the ResNet stem.  <3, 1>: VGG16 conv1_1 / the HED trunk's first conv (16 x 32 output tiles; a kernel row is ONE k-step: taps 0, 1 | 2, -).
template <int KS, int S>
struct PairCfg {
    static constexpr int RPW = S == 1 ? 4 : 2, TH = RPW * NWAVE;
    static constexpr int HH = (TH - 1) * S + KS, HW = 31 * S + KS + 1, PIX = HH * HW;        // (+ one column for the pair words)
    static constexpr int ROUNDS = (PIX + NT - 1) / NT;
    static constexpr int HPR = (KS + 3) / 4, NKS = KS * HPR;                                 // k-steps per kernel row, in all
    static constexpr int WBYTES = NKS * 2 * 1024, HBYTES = (PIX * 16 + 1023) / 1024 * 1024;
    static constexpr int LDS = WBYTES + HBYTES + NWAVE * PATCH_BYTES;
};

template <int KS, int S>
__global__ __launch_bounds__(NT, 2) void conv_stem_pair_kernel(const ConvLaunch d, const StemPairArgs a, const int ntiles) {
    using C = PairCfg<KS, S>;
    constexpr int RPW = C::RPW, TH = C::TH, HW = C::HW, PAD = KS / 2;
    constexpr int SP_ROUNDS = C::ROUNDS, SP_PIX = C::PIX, SP_NKS = C::NKS, SP_WBYTES = C::WBYTES, SP_HBYTES = C::HBYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wlds = smem;
    char* hbuf = smem + SP_WBYTES;
    char* pbuf = hbuf + SP_HBYTES;                // transpose patches (disjoint from the halo: the next halo is written while waves may still store)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int tiles_x = (d.OW + TW - 1) / TW, tiles_y = (d.OH + TH - 1) / TH, tpi = tiles_x * tiles_y;
    const int per_xcd = (ntiles + 7) >> 3, SS = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int span_lo = xcd * per_xcd, span_hi = min(span_lo + per_xcd, ntiles);
    int tile = span_lo + slot;
    if (tile >= span_hi) return;
    for (int i = tid; i < SP_WBYTES / 16; i += NT) *(float4*)(wlds + i * 16) = *(const float4*)((const char*)d.w_frag + i * 16);
    for (int i = tid; i < SP_HBYTES / 16; i += NT) *(float4*)(hbuf + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

    const long plane = (long)d.H * d.W;
    float pv[SP_ROUNDS][3];       // raw loads: nothing may consume them before store_halo (a use right behind its load is a full-latency wait per load)
    unsigned pin = 0;             // bit j: the pixel of round j lies inside the image
    auto load_halo = [&](int t) {
        const int n = t / tpi, r = t - n * tpi;
        const int y0 = (r / tiles_x) * TH, x0 = (r % tiles_x) * TW;
        const float* img = a.x + (long)n * a.C * plane;
        const float* pl[3] = {img + (long)a.perm[0] * plane, img + (long)a.perm[1] * plane, img + (long)a.perm[2] * plane};      // uniform bases: the loads take the
#pragma unroll                                                                                                              //  scalar-base + 32-bit lane offset form
        for (int j = 0; j < SP_ROUNDS; ++j) {
            const int p = min(j * NT + tid, SP_PIX - 1);
            const int hy = p / HW, hx = p - hy * HW;
            const int iy = y0 * S - PAD + hy, ix = x0 * S - PAD + hx;
            const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
            const unsigned off = (unsigned)(min(max(iy, 0), d.H - 1) * d.W + min(max(ix, 0), d.W - 1));
            pin = j == 0 ? (unsigned)inb : pin | ((unsigned)inb << j);
#pragma unroll
            for (int c = 0; c < 3; ++c) pv[j][c] = pl[c][off];
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int j = 0; j < SP_ROUNDS; ++j) {
            const int p = j * NT + tid;
            if (p < SP_PIX) {
                const int hy = p / HW, hx = p - hy * HW;
                const bool inb = (pin >> j) & 1u;
                float v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (inb && c < a.C) ? pv[j][c] * a.scale[c] + a.shift[c] : 0.f;       // zero padding is applied to the TRANSFORMED image
                f16x4 h; h[0] = (f16)v[0]; h[1] = (f16)v[1]; h[2] = (f16)v[2]; h[3] = (f16)0.f;
                *(f16x4*)(hbuf + p * 16) = h;
                if (hx > 0) *(f16x4*)(hbuf + p * 16 - 8) = h;
            }
        }
    };
    const int a_lane = ((wave * RPW * S) * HW + fr * S + 2 * fh) * 16;
    float4 bvs[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) bvs[j][g] = d.bias ? *(const float4*)(d.bias + j * 32 + 8 * g + 4 * fh) : make_float4(0.f, 0.f, 0.f, 0.f);
    load_halo(tile);
    __syncthreads();                                   // weights + zeroed buffer
    for (;;) {
        const int n = tile / tpi, r = tile - n * tpi;
        const int y0 = (r / tiles_x) * TH, x0 = (r % tiles_x) * TW;
        store_halo();
        __syncthreads();
        const int nxt = tile + SS;
        if (nxt < span_hi) load_halo(nxt);             // lands under the MFMAs and the epilogue

        f32x16 acc[RPW][2];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        {
            constexpr int PF = SP_NKS >= 3 ? 3 : SP_NKS;
            f16x8 af[PF][RPW], bf[PF][2];
            auto frags = [&](int ks, f16x8 (&av)[RPW], f16x8 (&bv)[2]) {
                const int ty = ks / C::HPR, h = ks % C::HPR;
#pragma unroll
                for (int i = 0; i < RPW; ++i) av[i] = *(const f16x8*)(hbuf + a_lane + ((i * S + ty) * HW + 4 * h) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = *(const f16x8*)(wlds + ((ks * 2 + j) * 64 + lane) * 16);
            };
#pragma unroll
            for (int p = 0; p < PF - 1; ++p) frags(p, af[p], bf[p]);
#pragma unroll
            for (int ks = 0; ks < SP_NKS; ++ks) {
                if (ks + PF - 1 < SP_NKS) frags(ks + PF - 1, af[(ks + PF - 1) % PF], bf[(ks + PF - 1) % PF]);
#pragma unroll
                for (int i = 0; i < RPW; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks % PF][j], af[ks % PF][i], acc[i][j], 0, 0, 0);
            }
        }
        // ---- epilogue: per 32-pixel row block through the wave's private patch (as the fp16 stem)
        f16* patch = (f16*)(pbuf + wave * PATCH_BYTES);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = j * 32 + 8 * g + 4 * fh;
                    const float4 bv = bvs[j][g];
                    float v0 = acc[i][j][4 * g] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y, v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
                    if (d.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    f16x4 h; h[0] = (f16)v0; h[1] = (f16)v1; h[2] = (f16)v2; h[3] = (f16)v3;
                    *(f16x4*)(patch + fr * CP + col) = h;
                }
            const int oy = y0 + wave * RPW + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane + 64 * q, px = idx >> 3, ch = idx & 7;
                const f16x8 v = *(const f16x8*)(patch + px * CP + ch * 8);
                if ((oy < d.OH) & (x0 + px < d.OW)) *(f16x8*)(d.out + ((long)((n * d.OH + oy) * d.OW + x0 + px) * 64 + ch * 8)) = v;
            }
        }
        __syncthreads();                               // every wave is done reading the halo
        tile = nxt;
        if (tile >= span_hi) break;
    }
}

// ... and with the MaxPool2d(3, stride 2, pad 1) that follows the ResNet stem fused in: the 1.07 GB stem output of a 32 x 1024^2 batch is never written.
// A workgroup owns 7 x 15 POOLED pixels = conv rows 2 py0 - 1 .. + 15 (four per wave) x conv columns 2 px0 - 1 .. + 31: neighbouring tiles recompute
// one conv row / column of overlap (16 / 14 x 32 / 30 = 1.22x the MFMAs, which are not the bound).  After bias + ReLU (conv samples outside the conv
// output count as 0 -- equivalent to the pool's -inf padding behind a ReLU): max over 3 columns by two wave-shift DPP steps in the accumulator
// registers (lane = conv column), max over 3 rows inside the wave, plus ONE row fetched from the next wave through LDS for the wave's second
// pooled row; even-column lanes then go through the wave's transpose patch and leave as 128-byte lines.  (Rounding to fp16 commutes with max.)
#ifndef GDT_STEM_POOL_EARLY_PREFETCH
#define GDT_STEM_POOL_EARLY_PREFETCH 0      // 1: next tile's loads before the MFMAs (14 spilled registers; measured 0.58 vs 0.55 ms)
#endif
constexpr int SP_HW = PairCfg<7, 2>::HW, SP_NKS = PairCfg<7, 2>::NKS, SP_WBYTES = PairCfg<7, 2>::WBYTES;
constexpr int SQ_HH = 37, SQ_PIX = SQ_HH * SP_HW, SQ_ROUNDS = (SQ_PIX + NT - 1) / NT;      // halo of 16 x 32 conv outputs; 11 loader rounds
constexpr int SQ_HBYTES = (SQ_PIX * 16 + 1023) / 1024 * 1024;
constexpr int SQ_LDS = SP_WBYTES + SQ_HBYTES + 256;                                          // + the bias vector
static_assert(SQ_HBYTES >= NWAVE * 2048 + NWAVE * 16 * CP * 2, "exchange rows and patches alias the consumed halo");

__global__ __launch_bounds__(NT, 2) void conv_stem_pair_pool_kernel(const ConvLaunch d, const StemPairArgs a, const int ntiles, const int PH, const int PW) {
    constexpr int RPW = 4, HW = SP_HW, PR = 7, PC = 15;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wlds = smem;
    char* hbuf = smem + SP_WBYTES;
    float* blds = (float*)(hbuf + SQ_HBYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int tiles_x = (PW + PC - 1) / PC, tiles_y = (PH + PR - 1) / PR, tpi = tiles_x * tiles_y;
    const int per_xcd = (ntiles + 7) >> 3, SS = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int span_lo = xcd * per_xcd, span_hi = min(span_lo + per_xcd, ntiles);
    int tile = span_lo + slot;
    if (tile >= span_hi) return;
    for (int i = tid; i < SP_WBYTES / 16; i += NT) *(float4*)(wlds + i * 16) = *(const float4*)((const char*)d.w_frag + i * 16);
    if (tid < 64) blds[tid] = d.bias ? d.bias[tid] : 0.f;

    const long plane = (long)d.H * d.W;
    float pv[SQ_ROUNDS][3];
    unsigned pin = 0;
    auto load_halo = [&](int t) {
        const int n = t / tpi, r = t - n * tpi;
        const int cy0 = (r / tiles_x) * (2 * PR) - 1, cx0 = (r % tiles_x) * (2 * PC) - 1;
        const float* img = a.x + (long)n * a.C * plane;
        const float* pl[3] = {img + (long)a.perm[0] * plane, img + (long)a.perm[1] * plane, img + (long)a.perm[2] * plane};      // uniform bases: the loads take the
#pragma unroll                                                                                                              //  scalar-base + 32-bit lane offset form
        for (int j = 0; j < SQ_ROUNDS; ++j) {
            const int p = min(j * NT + tid, SQ_PIX - 1);
            const int hy = p / HW, hx = p - hy * HW;
            const int iy = cy0 * 2 - 3 + hy, ix = cx0 * 2 - 3 + hx;
            const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
            const unsigned off = (unsigned)(min(max(iy, 0), d.H - 1) * d.W + min(max(ix, 0), d.W - 1));
            pin = j == 0 ? (unsigned)inb : pin | ((unsigned)inb << j);
#pragma unroll
            for (int c = 0; c < 3; ++c) pv[j][c] = pl[c][off];
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int j = 0; j < SQ_ROUNDS; ++j) {
            const int p = j * NT + tid;
            if (p < SQ_PIX) {
                const int hy = p / HW, hx = p - hy * HW;
                const bool inb = (pin >> j) & 1u;
                float v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = (inb && c < a.C) ? pv[j][c] * a.scale[c] + a.shift[c] : 0.f;
                f16x4 h; h[0] = (f16)v[0]; h[1] = (f16)v[1]; h[2] = (f16)v[2]; h[3] = (f16)0.f;
                *(f16x4*)(hbuf + p * 16) = h;
                // the high half of the left neighbour's word; the last word of a row keeps a (finite) stale high half: it only meets tap 7's zero weights
                if (hx > 0) *(f16x4*)(hbuf + p * 16 - 8) = h;
            }
        }
    };
    const int a_lane = ((wave * RPW * 2) * HW + fr * 2 + 2 * fh) * 16;
    char* xrow = hbuf;                                     // [wave][16 pooled columns][64 channels] fp16: every wave's first conv row after the column max
    f16* patch = (f16*)(hbuf + NWAVE * 2048) + wave * (16 * CP);
    for (int i = tid; i < SQ_HBYTES / 16; i += NT) *(float4*)(hbuf + i * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    load_halo(tile);
    __syncthreads();
    for (;;) {
        const int n = tile / tpi, r = tile - n * tpi;
        const int py0 = (r / tiles_x) * PR, px0 = (r % tiles_x) * PC;
        const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;
        store_halo();
        __syncthreads();
        const int nxt = tile + SS;
#if GDT_STEM_POOL_EARLY_PREFETCH
        if (nxt < span_hi) load_halo(nxt);
#endif

        f32x16 acc[RPW][2];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {          // the accumulators start from the bias: register 4g + e of lane (fr, fh) is channel j * 32 + 8g + 4fh + e
                    const float4 bv = *(const float4*)(blds + j * 32 + 8 * g + 4 * fh);
                    acc[i][j][4 * g] = bv.x; acc[i][j][4 * g + 1] = bv.y; acc[i][j][4 * g + 2] = bv.z; acc[i][j][4 * g + 3] = bv.w;
                }
        {
            constexpr int PF = 2;
            f16x8 af[PF][RPW], bf[PF][2];
            auto frags = [&](int ks, f16x8 (&av)[RPW], f16x8 (&bv)[2]) {
                const int ty = ks >> 1, h = ks & 1;
#pragma unroll
                for (int i = 0; i < RPW; ++i) av[i] = *(const f16x8*)(hbuf + a_lane + ((i * 2 + ty) * HW + 4 * h) * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) bv[j] = *(const f16x8*)(wlds + ((ks * 2 + j) * 64 + lane) * 16);
            };
            frags(0, af[0], bf[0]);
#pragma unroll
            for (int ks = 0; ks < SP_NKS; ++ks) {
                if (ks + 1 < SP_NKS) frags(ks + 1, af[(ks + 1) % PF], bf[(ks + 1) % PF]);
#pragma unroll
                for (int i = 0; i < RPW; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[ks % PF][j], af[ks % PF][i], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                               // every wave is done reading the halo: its bytes now hold the exchange rows and the patches
        // ---- ReLU, validity mask, rounding to fp16 pairs (64 registers instead of 128; rounding commutes with max), then the column max on the pairs:
        // lane fr = conv column cx0 + fr; two wave-shift steps give max over columns fr, fr + 1, fr + 2 (used at even fr <= 28)
        typedef f16 f16x2 __attribute__((ext_vector_type(2)));
        f16x2 hq[RPW][2][8];
        {
            const bool col_ok = (unsigned)(cx0 + fr) < (unsigned)d.OW;
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const bool ok = col_ok & ((unsigned)(cy0 + wave * RPW + i) < (unsigned)d.OH);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        // one packed convert, ReLU and mask on the pair (three instructions per two values).  (Not inline asm: the hazard
                        // recogniser does not see an asm statement's read of a register the MFMAs just before it are still writing.)
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        const f32x2 pr2 = {acc[i][j][2 * k], acc[i][j][2 * k + 1]};
                        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(pr2, f16x2));
                        f16x2 h = __builtin_bit_cast(f16x2, ok ? pk : 0u);
                        const f16x2 zero2 = {(f16)0.f, (f16)0.f};
                        if (d.relu) h = __builtin_elementwise_max(h, zero2);
                        hq[i][j][k] = h;
                    }
            }
        }
#if !GDT_STEM_POOL_EARLY_PREFETCH
        if (nxt < span_hi) load_halo(nxt);             // (here, not before the MFMAs: the accumulators are dead now and the 33 halo registers fit; the rest of the epilogue covers the latency)
#endif
        auto shl1h = [](f16x2 v) -> f16x2 { return __builtin_bit_cast(f16x2, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false)); };
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f16x2 t = __builtin_elementwise_max(hq[i][j][k], shl1h(hq[i][j][k]));
                    hq[i][j][k] = __builtin_elementwise_max(t, shl1h(t));
                }
        // ---- the wave's first row goes to LDS for the wave above it (pair 2g, 2g + 1 of block j = channels j * 32 + 8g + 4fh .. + 3)
        const bool writer = (fr & 1) == 0;
        const int pc = fr >> 1;
        auto quad = [](f16x2 lo, f16x2 hi) -> uint2 { return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)); };
        if (writer) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) *(uint2*)(xrow + wave * 2048 + (pc * 64 + j * 32 + 8 * g + 4 * fh) * 2) = quad(hq[0][j][2 * g], hq[0][j][2 * g + 1]);
        }
        __syncthreads();
        // ---- row max + store: pooled rows 2 wave (conv rows 0, 1, 2 of the wave) and 2 wave + 1 (rows 2, 3 and row 0 of the next wave; none for the last wave)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
            const int pr = 2 * wave + sidx;
            const int py = py0 + pr;
            const bool row_ok = (pr < PR) & (py < PH);       // (wave-uniform)
            if (writer) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f16x2 m0, m1;
                        if (sidx == 0) {
                            m0 = __builtin_elementwise_max(__builtin_elementwise_max(hq[0][j][2 * g], hq[1][j][2 * g]), hq[2][j][2 * g]);
                            m1 = __builtin_elementwise_max(__builtin_elementwise_max(hq[0][j][2 * g + 1], hq[1][j][2 * g + 1]), hq[2][j][2 * g + 1]);
                        } else {
                            const uint2 nx = *(const uint2*)(xrow + min(wave + 1, NWAVE - 1) * 2048 + (pc * 64 + j * 32 + 8 * g + 4 * fh) * 2);
                            m0 = __builtin_elementwise_max(__builtin_elementwise_max(hq[2][j][2 * g], hq[3][j][2 * g]), __builtin_bit_cast(f16x2, nx.x));
                            m1 = __builtin_elementwise_max(__builtin_elementwise_max(hq[2][j][2 * g + 1], hq[3][j][2 * g + 1]), __builtin_bit_cast(f16x2, nx.y));
                        }
                        *(uint2*)(patch + pc * CP + j * 32 + 8 * g + 4 * fh) = quad(m0, m1);
                    }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = lane + 64 * q, px = idx >> 3, ch = idx & 7;
                const f16x8 v = *(const f16x8*)(patch + px * CP + ch * 8);
                if (row_ok & (px < PC) & (px0 + px < PW)) *(f16x8*)(d.out + ((long)((n * PH + py) * PW + px0 + px) * 64 + ch * 8)) = v;
            }
        }
        __syncthreads();                               // exchange rows / patches are dead: the next halo may be written
        tile = nxt;
        if (tile >= span_hi) break;
    }
}

}  // namespace

// 8-channel (image) input, exactly 64 output channels, fp16 NHWC output, 7x7 (stride 1 or 2, pad 3) or 3x3 (stride 1, pad 1);
// fused statistics need whole 128-pixel records per wave: stride 1 and OW % 32 == 0, OH % 16 == 0.
bool gdt_conv_stem_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_STEM"); return e ? atoi(e) : 1; }();
    if (mode == 0 || !d.w_frag || d.Cin != 8 || d.Cout != 64 || d.CoutPad != 64 || d.out_f32 || !d.out || d.res || d.in_norm || d.pool2) return false;
    if (d.sy != d.sx || d.dys != 1 || d.dxs != 1 || d.osy != 1 || d.osx != 1 || d.ooy != 0 || d.oox != 0) return false;
    const bool k7 = d.ntaps == 49 && d.TW == 7 && d.dy0 == -3 && d.dx0 == -3 && (d.sy == 1 || d.sy == 2);
    const bool k3 = d.ntaps == 9 && d.TW == 3 && d.dy0 == -1 && d.dx0 == -1 && d.sy == 1;
    if (!k7 && !k3) return false;
    if (d.pad_reflect && (d.H <= 3 || d.W <= 3)) return false;
    if (d.stats && (d.sy != 1 || d.OW % 32 != 0 || d.OH % 16 != 0)) return false;
    // (pixel indices are ints, element offsets longs: 32 x 3 x 1024 x 1024 -> 2^31 output elements is fine)
    return (long)d.N * d.OH * d.OW < (1L << 31) && (long)d.N * d.H * d.W < (1L << 31) && (long)d.N * d.OH * d.OW >= 65536;
}

int gdt_launch_conv_stem(const ConvLaunch& d, hipStream_t stream) {
    if (d.ntaps == 9) return launch_stem<3, 1>(d, stream);
    return d.sy == 1 ? launch_stem<7, 1>(d, stream) : launch_stem<7, 2>(d, stream);
}

// "f16c" form (AUG): the input tensor holds augmented fp16 pixel words (gdt_k_pack_input, aug = 1), weights W1 (augmented, w_frag) and W2
// (residuals, w_frag2) in the stem fragment order; fp32 NHWC output.  Same shapes as the fp16 form.
bool gdt_conv_stem_c_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_STEM"); return e ? atoi(e) : 1; }();
    if (mode == 0 || !d.w_frag || !d.w_frag2 || d.Cin != 8 || d.Cout != 64 || d.CoutPad != 64 || d.out_f32 || !d.out || d.res || d.in_norm || d.pool2) return false;
    if (d.sy != d.sx || d.dys != 1 || d.dxs != 1 || d.osy != 1 || d.osx != 1 || d.ooy != 0 || d.oox != 0) return false;
    const bool k7 = d.ntaps == 49 && d.TW == 7 && d.dy0 == -3 && d.dx0 == -3 && (d.sy == 1 || d.sy == 2);
    const bool k3 = d.ntaps == 9 && d.TW == 3 && d.dy0 == -1 && d.dx0 == -1 && d.sy == 1;
    if (!k7 && !k3) return false;
    if (d.pad_reflect && (d.H <= 3 || d.W <= 3)) return false;
    if (d.stats && (d.sy != 1 || d.OW % 32 != 0 || d.OH % 16 != 0)) return false;
    return (long)d.N * d.OH * d.OW * 64 < (1L << 32) && (long)d.N * d.H * d.W < (1L << 31) && (long)d.N * d.OH * d.OW >= 65536;      // (unsigned 32-bit element offsets in the fp32 epilogue)
}

int gdt_launch_conv_stem_c(const ConvLaunch& d, hipStream_t stream) {
    if (d.ntaps == 9) return launch_stem<3, 1, true>(d, stream);
    return d.sy == 1 ? launch_stem<7, 1, true>(d, stream) : launch_stem<7, 2, true>(d, stream);
}


// Direct form of the ResNet stem (see conv_stem_pair_kernel): the descriptor is the stem conv's (H, W = the image, w_frag = the pair-packed weights of
// net.hip, fp16 NHWC output), x the caller's fp32 NCHW image with at most 3 channels, perm / scale / shift the input op's per-channel transform.
bool gdt_conv_stem_pair_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_STEM_PAIR"); return e ? atoi(e) : 1; }();
    if (mode == 0 || !d.w_frag || d.Cout != 64 || d.CoutPad != 64 || d.out_f32 || !d.out || d.res || d.in_norm || d.pool2 || d.stats || d.pad_reflect) return false;
    const bool k7s2 = d.ntaps == 49 && d.TW == 7 && d.dy0 == -3 && d.dx0 == -3 && d.sy == 2 && d.sx == 2;
    const bool k3s1 = d.ntaps == 9 && d.TW == 3 && d.dy0 == -1 && d.dx0 == -1 && d.sy == 1 && d.sx == 1;
    if ((!k7s2 && !k3s1) || d.dys != 1 || d.dxs != 1) return false;
    if (d.osy != 1 || d.osx != 1 || d.ooy != 0 || d.oox != 0 || d.H < 8 || d.W < 8) return false;
    return (long)d.N * d.OH * d.OW < (1L << 31) && (long)d.H * d.W < (1L << 31) && (long)d.N * d.OH * d.OW >= 65536;
}

template <int KS, int S>
static int launch_stem_pair(const ConvLaunch& d, const StemPairArgs& a, hipStream_t stream) {
    using C = PairCfg<KS, S>;
    static_assert(2 * C::LDS <= 160 * 1024, "two workgroups per CU");
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_stem_pair_kernel<KS, S>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    const int ntiles = d.N * ((d.OW + TW - 1) / TW) * ((d.OH + C::TH - 1) / C::TH);
    const int grid = min(2 * cus, (ntiles + 7) / 8 * 8);
    hipLaunchKernelGGL((conv_stem_pair_kernel<KS, S>), dim3(grid), dim3(NT), C::LDS, stream, d, a, ntiles);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_launch_conv_stem_pair(const ConvLaunch& d, const float* x, int C, const int* perm, const float* scale, const float* shift, hipStream_t stream) {
    GDT_REQUIRE(x != nullptr && C >= 1 && C <= 3, "stem: 1..3 image channels");
    StemPairArgs a;
    a.x = x; a.C = C;
    for (int c = 0; c < 4; ++c) { a.perm[c] = c < C ? perm[c] : 0; a.scale[c] = c < C ? scale[c] : 0.f; a.shift[c] = c < C ? shift[c] : 0.f; }
    return d.ntaps == 9 ? launch_stem_pair<3, 1>(d, a, stream) : launch_stem_pair<7, 2>(d, a, stream);
}

// ... with the following MaxPool2d(3, 2, 1) fused: d.out is the POOLED tensor [N][PH][PW][64]
int gdt_launch_conv_stem_pair_pool(const ConvLaunch& d, const float* x, int C, const int* perm, const float* scale, const float* shift, int PH, int PW,
                                   hipStream_t stream) {
    GDT_REQUIRE(x != nullptr && C >= 1 && C <= 3, "stem: 1..3 image channels");
    GDT_REQUIRE(PH == (d.OH - 1) / 2 + 1 && PW == (d.OW - 1) / 2 + 1, "stem: pooled size");
    static_assert(2 * SQ_LDS <= 160 * 1024, "two workgroups per CU");
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_stem_pair_pool_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SQ_LDS));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    StemPairArgs a;
    a.x = x; a.C = C;
    for (int c = 0; c < 4; ++c) { a.perm[c] = c < C ? perm[c] : 0; a.scale[c] = c < C ? scale[c] : 0.f; a.shift[c] = c < C ? shift[c] : 0.f; }
    const int ntiles = d.N * ((PW + 14) / 15) * ((PH + 6) / 7);
    const int grid = min(2 * cus, (ntiles + 7) / 8 * 8);
    hipLaunchKernelGGL(conv_stem_pair_pool_kernel, dim3(grid), dim3(NT), SQ_LDS, stream, d, a, ntiles, PH, PW);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

