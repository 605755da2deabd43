// "f16x3" implicit-GEMM convolution: fp32 NHWC activations, fp32-accurate results on the fp16 matrix cores.
//
// Why: a single fp16 MFMA pass rounds activations and weights to 11 bits; through the 24 conv layers of the generator on
// random weights that accumulates to ~2.5e-3, above north_star's 1e-3.  Here every operand is split
// into two fp16 numbers, x = hi + lo * 2^-11 (hi = fp16(x), lo = fp16((x - hi) * 2^11)), and three MFMA passes are
// accumulated in fp32:   acc += a_hi*b_hi;   acc_lo += a_lo*b_hi + a_hi*b_lo;   result = acc + acc_lo * 2^-11
// (the lo*lo term is 2^-22 relative and dropped).  Products of fp16 values are exact in the fp32 accumulator, so the result
// carries ~22 mantissa bits per operand: fp32-class accuracy at 3x the MFMA work, 1/5 the cost of the f32 MFMA path.
//
// Same GEMM view, tap table, padding and tile mapping as conv_igemm.hip.  Differences: K-step 32; the A operand is read as
// fp32 through registers (global_load_dwordx4), split, and written to LDS as two fp16 images (issue-early / write-late
// staging); the weights are pre-split on the host and staged with global_load_lds; the epilogue stores fp32 NHWC straight
// from the accumulators (one half-wave writes 128 contiguous bytes).
#include <cstdlib>

#include "gdt_common.h"

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int BK = 32;
constexpr int ROWB = 64;                 // bytes per LDS row (32 halves)
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_dst, 16, 0, 0);
}

template <int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_x3_kernel(const ConvLaunch d) {
    static_assert(WGM * WGN == 4, "4 wavefronts");
    constexpr int BM = 128;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE = 2 * A_BYTES + 2 * B_BYTES;          // A_hi, A_lo, B_hi, B_lo
    constexpr int BR = (BN + 63) / 64;                        // weight staging rounds per matrix (64 rows per round)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const float* __restrict__ in = (const float*)d.in;

    const int ntn = d.CoutPad / BN;
    int tile_m, tile_n;
    if (!gdt_tile_of_block(blockIdx.x, (d.M + BM - 1) / BM, ntn, tile_m, tile_n)) return;

    // ---- A staging state: this thread owns 4-channel group c4 = tid & 7 of rows (tid >> 3) + 32*i
    const int c4 = tid & 7, arow = tid >> 3;
    const int hw_g = d.OHg * d.OWg;
    int a_base[4], a_iy0[4], a_ix0[4];
    unsigned a_valid = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = tile_m * BM + r * 32 + arow;
        const int mm = m < d.M ? m : 0;
        const int n = mm / hw_g, rem = mm - n * hw_g;
        const int oy = rem / d.OWg, ox = rem - oy * d.OWg;
        a_base[r] = n * d.H * d.W; a_iy0[r] = oy * d.sy; a_ix0[r] = ox * d.sx;
        a_valid |= (m < d.M ? 1u : 0u) << r;
    }
    const bool refl = d.pad_reflect != 0;
    const int g4mask = (2 << d.lc8) - 1;                       // Cin / 4 - 1
    float4 areg[4];
    // the producer's InstanceNorm (+ReLU) applied while staging (round 5: the stride-2 / transposed layers of the generator; a tile lies inside one image --
    // gdt_conv_x3_norm_eligible): (mean, rstd) of the step's 4 channels travel with the pieces and are applied at the LDS write; padding stays zero
    const float* __restrict__ nrm = d.in_norm ? d.in_norm + (long)((tile_m * BM) / hw_g) * d.Cin * 2 : nullptr;
    float4 nm0 = make_float4(0.f, 1.f, 0.f, 1.f), nm1 = nm0;
    unsigned a_okmask = 0;
    auto load_a = [&](int ks) {
        const int g4 = ks * 8 + c4;
        const int tap = g4 >> (d.lc8 + 1), coff = (g4 & g4mask) * 4;
        if (nrm) { nm0 = *(const float4*)(nrm + coff * 2); nm1 = *(const float4*)(nrm + coff * 2 + 4); a_okmask = 0; }
        const int ty = (tap * d.invTW) >> 16, tx = tap - ty * d.TW;
        const int dy = d.dy0 + ty * d.dys, dx = d.dx0 + tx * d.dxs;
        const bool tap_ok = tap < d.ntaps;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = a_iy0[r] + dy, ix = a_ix0[r] + dx;
            const int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
            const int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
            const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
            const bool ok = tap_ok & (((a_valid >> r) & 1u) != 0) & (inb | refl);
            const int pix = a_base[r] + ry * d.W + rx;
            const float* src = in + (((long)pix << (d.lc8 + 3)) + coff);
            areg[r] = ok ? *(const float4*)src : make_float4(0.f, 0.f, 0.f, 0.f);
            a_okmask |= (ok ? 1u : 0u) << r;
        }
    };
    auto store_a = [&](int stage) {
        char* Ah = smem + stage * STAGE;
        char* Al = Ah + A_BYTES;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r * 32 + arow;
            const int off = row * ROWB + (((c4 >> 1) ^ ((row >> 2) & 3)) << 4) + (c4 & 1) * 8;
            float x[4] = {areg[r].x, areg[r].y, areg[r].z, areg[r].w};
            if (nrm && ((a_okmask >> r) & 1u)) {
                x[0] = (x[0] - nm0.x) * nm0.y; x[1] = (x[1] - nm0.z) * nm0.w; x[2] = (x[2] - nm1.x) * nm1.y; x[3] = (x[3] - nm1.z) * nm1.w;
                if (d.in_relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
            }
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = (f16)x[e];
                lo[e] = (f16)((x[e] - (float)hi[e]) * LO_SCALE);
            }
            *(f16x4*)(Ah + off) = hi;
            *(f16x4*)(Al + off) = lo;
        }
    };
    // ---- B staging (pre-split weights, global_load_lds): lane stages chunk (tid & 3) of rows (tid >> 2) + 64*r
    const int brow = tid >> 2;
    const int bq = (tid & 3) ^ ((brow >> 2) & 3);
    const f16* bh_src = d.w + ((long)(tile_n * BN + brow) * d.Kpad + bq * 8);
    const f16* bl_src = d.w_lo + ((long)(tile_n * BN + brow) * d.Kpad + bq * 8);
    auto issue_b = [&](int ks, int stage) {
        char* Bh = smem + stage * STAGE + 2 * A_BYTES;
        char* Bl = Bh + B_BYTES;
#pragma unroll
        for (int r = 0; r < BR; ++r) {
            if (r * 64 + wave * 16 >= BN) continue;            // wave-uniform (BN = 32: waves 0, 1 only)
            glds16(bh_src + ((long)r * 64 * d.Kpad + ks * BK), Bh + (r * 64 + wave * 16) * ROWB);
            glds16(bl_src + ((long)r * 64 * d.Kpad + ks * BK), Bl + (r * 64 + wave * 16) * ROWB);
        }
    };

    f32x16 acc[TM][TN], accl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accl[i][j][e] = 0.f; }

    const int fr = lane & 31, fh = lane >> 5;
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * WTM + i * 32 + fr; a_off[i] = row * ROWB; a_sw[i] = (row >> 2) & 3; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int row = wn * WTN + j * 32 + fr; b_off[j] = row * ROWB; b_sw[j] = (row >> 2) & 3; }

    load_a(0);
    issue_b(0, 0);
    store_a(0);
    for (int ks = 0; ks < d.nk; ++ks) {
        __syncthreads();      // stage ks complete (A written, weight DMA landed); the other stage is free
        const bool more = ks + 1 < d.nk;
        if (more) { load_a(ks + 1); issue_b(ks + 1, (ks + 1) & 1); }      // in flight while this step computes
        const char* Ah = smem + (ks & 1) * STAGE;
        const char* Al = Ah + A_BYTES;
        const char* Bh = Ah + 2 * A_BYTES;
        const char* Bl = Bh + B_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = 2 * kk + fh;
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int o = a_off[i] + ((ch ^ a_sw[i]) << 4);
                ah[i] = *(const f16x8*)(Ah + o); al[i] = *(const f16x8*)(Al + o);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int o = b_off[j] + ((ch ^ b_sw[j]) << 4);
                bh[j] = *(const f16x8*)(Bh + o); bl[j] = *(const f16x8*)(Bl + o);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accl[i][j], 0, 0, 0);
                    accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accl[i][j], 0, 0, 0);
                }
        }
        if (more) store_a((ks + 1) & 1);     // write-late: the loads had the whole MFMA phase to land
    }

    // ---------------------------------------------------------------- epilogue: fp32 straight from the accumulators
    const int ohw = d.OH * d.OW;
    float* outp = (float*)d.out;
    const float* resp = (const float*)d.res;
    float* sl = (float*)smem;                 // [WGM][BN][2] statistics scratch (staging memory is free now)
    if (d.stats) __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int lcol = wn * WTN + j * 32 + fr;
        const int col = tile_n * BN + lcol;
        const float bv = (d.bias && col < d.CoutPad) ? d.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        float rr[TM][16];                      // residual values of this column: all loads issued before any is consumed
        if (resp) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    const int m = tile_m * BM + row;
                    const bool ok = m < d.M && col < d.Cout;
                    const int mm = ok ? m : 0;
                    const int n = mm / hw_g, rem = mm - n * hw_g;
                    const int oy = rem / d.OWg, ox = rem - oy * d.OWg;
                    const long off = ((long)n * ohw + (long)(oy * d.osy + d.ooy) * d.OW + ox * d.osx + d.oox) * d.Cout + (ok ? col : 0);
                    rr[i][e] = resp[ok ? off : 0];
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                const int m = tile_m * BM + row;
                float v = acc[i][j][e] + accl[i][j][e] * LO_INV + bv;
                s1 += v; s2 += v * v;
                if (m >= d.M || col >= d.Cout) continue;
                const int n = m / hw_g, rem = m - n * hw_g;
                const int oy = rem / d.OWg, ox = rem - oy * d.OWg;
                const long opix = (long)(oy * d.osy + d.ooy) * d.OW + ox * d.osx + d.oox;
                if (d.out_f32) {
                    if (d.relu) v = fmaxf(v, 0.f);
                    if (d.act == 1) v = tanhf(v);
                    else if (d.act == 2) v = 1.f / (1.f + __expf(-v));
                    d.out_f32[((long)n * d.Cout + col) * ohw + opix] = v;
                } else {
                    const long off = ((long)n * ohw + opix) * d.Cout + col;
                    if (resp) v += rr[i][e];
                    if (d.relu) v = fmaxf(v, 0.f);
                    outp[off] = v;
                }
            }
        if (d.stats) {
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (fh == 0) { sl[(wm * BN + lcol) * 2 + 0] = s1; sl[(wm * BN + lcol) * 2 + 1] = s2; }
        }
    }
    if (d.stats) {
        __syncthreads();
        if (tid < BN) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WGM; ++w) { s1 += sl[(w * BN + tid) * 2 + 0]; s2 += sl[(w * BN + tid) * 2 + 1]; }
            const int gcol = tile_n * BN + tid;
            if (gcol < d.Cout) {
                float* dst = d.stats + ((long)(d.stats_tile_base + tile_m) * 2) * d.Cout + gcol;
                dst[0] = s1; dst[d.Cout] = s2;
            }
        }
    }
}

template <int BN, int WGM, int WGN>
int launch_x3(const ConvLaunch& d, hipStream_t stream) {
    const int ntm = (d.M + 127) / 128, ntn = d.CoutPad / BN;
    constexpr size_t lds = 2 * (size_t)(2 * 128 * ROWB + 2 * BN * ROWB);
    static_assert(lds <= 64 * 1024 && (size_t)WGM * BN * 8 <= lds, "LDS budget");
    hipLaunchKernelGGL((conv_igemm_x3_kernel<BN, WGM, WGN>), dim3(gdt_grid_for_tiles(ntm, ntn)), dim3(256), lds, stream, d);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// can the generic f16x3 GEMM apply the producer's InstanceNorm (+ReLU) while it stages?  (no residual, no write-back; a 128-row tile inside one image)
bool gdt_conv_x3_norm_eligible(const ConvLaunch& d) {
    return d.w_lo != nullptr && !d.in_res && !d.in_out && (d.OHg * d.OWg) % 128 == 0 && d.Cin % 4 == 0;
}

int gdt_launch_conv_x3(const ConvLaunch& d, hipStream_t stream, int* variant) {
    GDT_REQUIRE(d.Cin >= 8 && (d.Cin & (d.Cin - 1)) == 0 && (1 << d.lc8) * 8 == d.Cin, "Cin must be a power of two >= 8");
    GDT_REQUIRE(d.Kpad % 64 == 0 && d.nk == d.Kpad / BK && d.ntaps * d.Cin <= d.Kpad, "Kpad / nk (K-step 32)");
    GDT_REQUIRE(d.w_lo != nullptr && d.M > 0, "f16x3 needs the split weights");
    if (d.stats) GDT_REQUIRE(!d.out_f32 && !d.res && !d.relu && d.M % 128 == 0 && (d.OHg * d.OWg) % 128 == 0,
                             "fused InstanceNorm statistics need whole 128-row tiles per image and a plain conv epilogue");
    if (d.pad_reflect) {
        const int pady = d.dy0 < 0 ? -d.dy0 : 0, padx = d.dx0 < 0 ? -d.dx0 : 0;
        GDT_REQUIRE(pady < d.H && padx < d.W, "reflect padding needs pad < input size");
    }
    const int bn = gdt_conv_bn(d.Cout);
    GDT_REQUIRE(d.CoutPad % bn == 0 && d.CoutPad >= d.Cout, "CoutPad must be a multiple of the N tile");
    if (d.x3_form == 0 && gdt_conv_halo_x3_eligible(d)) { if (variant) *variant = 930128; return gdt_launch_conv_halo_x3(d, stream); }
    if (gdt_conv_halo_x3_taps_eligible(d)) { if (variant) *variant = d.x3_form == 2 ? 932128 : 931128; return gdt_launch_conv_halo_x3_taps(d, stream); }
    GDT_REQUIRE(d.x3_form == 0, "the space-to-depth view exists in the patch kernel only");
    GDT_REQUIRE(d.in_norm == nullptr || gdt_conv_x3_norm_eligible(d), "fused input normalisation: plain norm (+ReLU), whole 128-row tiles per image");
    if (variant) *variant = 300000 + bn;
    if (bn == 128) return launch_x3<128, 2, 2>(d, stream);
    if (bn == 64) return launch_x3<64, 2, 2>(d, stream);
    return launch_x3<32, 4, 1>(d, stream);
}
