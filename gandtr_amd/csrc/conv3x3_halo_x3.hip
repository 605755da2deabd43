// 3x3 / stride-1 / pad-1 convolution in the "f16x3" precision mode (fp32 NHWC activations, split-fp16 operands, three MFMA
// passes -- see conv_igemm_x3.hip) with the LDS-resident input halo of conv3x3_halo.hip.
//
// One workgroup = 16x16 output patch x 128 output channels, 8 wavefronts (4 x 2, 64x64 per wave, two accumulator sets).
// Per 32-channel chunk the 18x18 fp32 halo is read once through registers (global_load_dwordx4), optionally normalised
// (fused InstanceNorm + ReLU of the producer: x -> max((x - mean) * rstd, 0), p2p_networks.py:29,:272), split into
// hi / lo fp16 images and written to LDS; the 9 taps then run against it.  The pre-split weights stream per tap with
// global_load_lds.  Halo pieces of the next chunk are issued one per tap step and written one step later
// (issue-early / write-late), so their latency hides under a whole step of MFMAs.
#include <cstdlib>

#include "gdt_common.h"

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int ROWB = 64;                   // bytes per LDS row: 32 halves of K
constexpr int HALO_W = 18, HALO_ROWS = 324, HALO_ROWS_PAD = 328;
constexpr int A_BYTES = HALO_ROWS_PAD * ROWB;     // one of {hi, lo}
constexpr int BN = 128, B_BYTES = BN * ROWB;
constexpr int NT = 512;
constexpr int STAGE_A = 2 * A_BYTES, STAGE_B = 2 * B_BYTES;
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_dst, 16, 0, 0);
}

__global__ __launch_bounds__(NT) void conv3x3_halo_x3_kernel(const ConvLaunch d) {
    constexpr int WGN = 2, WTM = 64, WTN = 64, TM = 2, TN = 2;      // 4 x 2 wavefronts, 64 x 64 per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][A hi, A lo] [2][B hi, B lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const float* __restrict__ in = (const float*)d.in;

    const int tiles_x = (d.W + 15) >> 4, tiles_y = (d.H + 15) >> 4;
    const int tpi = tiles_x * tiles_y, ntm = d.N * tpi, ntn = d.CoutPad / BN;
    int tile_m, tile_n;
    if (!gdt_tile_of_block(blockIdx.x, ntm, ntn, tile_m, tile_n)) return;      // XCD-chunked, see gdt_common.h
    const int n = tile_m / tpi, tr = tile_m - n * tpi;
    const int y0 = (tr / tiles_x) << 4, x0 = (tr % tiles_x) << 4;

    // ---- halo staging state: this thread owns the 4-channel group c4 = tid & 7 of halo rows (tid >> 3) + 64*r
    const int c4 = tid & 7, hrow = tid >> 3;
    const bool refl = d.pad_reflect != 0;
    int a_pix[6]; unsigned a_ok = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int h = r * 64 + hrow;
        const int hy = h / HALO_W, hx = h - hy * HALO_W;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
        int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
        ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);
        const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
        a_pix[r] = (n * d.H + ry) * d.W + rx;
        a_ok |= ((h < HALO_ROWS) & (inb | refl) ? 1u : 0u) << r;
    }
    float4 nm0 = make_float4(0.f, 1.f, 0.f, 1.f), nm1 = nm0;        // (mean, rstd) x 4 channels of the chunk being staged
    auto load_norm = [&](int chunk) {
        if (!d.in_norm) return;
        const float* p = d.in_norm + ((long)n * d.Cin + chunk * 32 + c4 * 4) * 2;
        nm0 = *(const float4*)p; nm1 = *(const float4*)(p + 4);
    };
    auto load_piece = [&](int chunk, int r) -> float4 {
        if (r * 64 + hrow >= HALO_ROWS_PAD) return make_float4(0.f, 0.f, 0.f, 0.f);
        if (!((a_ok >> r) & 1u)) return make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v = *(const float4*)(in + (((long)a_pix[r] << (d.lc8 + 3)) + chunk * 32 + c4 * 4));
        if (d.in_norm) {
            v.x = (v.x - nm0.x) * nm0.y; v.y = (v.y - nm0.z) * nm0.w; v.z = (v.z - nm1.x) * nm1.y; v.w = (v.w - nm1.z) * nm1.w;
            if (d.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        return v;
    };
    auto store_piece = [&](int stage, int r, const float4& v) {
        const int row = r * 64 + hrow;
        if (row >= HALO_ROWS_PAD) return;
        char* Ah = smem + stage * STAGE_A;
        const int off = row * ROWB + (((c4 >> 1) ^ ((row >> 2) & 3)) << 4) + (c4 & 1) * 8;
        const float x[4] = {v.x, v.y, v.z, v.w};
        f16x4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (f16)x[e]; lo[e] = (f16)((x[e] - (float)hi[e]) * LO_SCALE); }
        *(f16x4*)(Ah + off) = hi;
        *(f16x4*)(Ah + A_BYTES + off) = lo;
    };
    // ---- weight staging: lane stages 16-byte chunk (tid & 3) of row tid >> 2 (128 rows), hi and lo
    const int brow = tid >> 2;
    const int bq = (tid & 3) ^ ((brow >> 2) & 3);
    const f16* bh_src = d.w + ((long)(tile_n * BN + brow) * d.Kpad + bq * 8);
    const f16* bl_src = d.w_lo + ((long)(tile_n * BN + brow) * d.Kpad + bq * 8);
    auto issue_b = [&](int koff, int stage) {
        char* Bh = smem + 2 * STAGE_A + stage * STAGE_B;
        glds16(bh_src + koff, Bh + (wave * 16) * ROWB);
        glds16(bl_src + koff, Bh + B_BYTES + (wave * 16) * ROWB);
    };

    f32x16 acc[TM][TN], accl[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accl[i][j][e] = 0.f; }

    const int fr = lane & 31, fh = lane >> 5;
    int a_h0[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int m = wm * WTM + i * 32 + fr; a_h0[i] = (m >> 4) * HALO_W + (m & 15); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int row = wn * WTN + j * 32 + fr; b_off[j] = row * ROWB; b_sw[j] = (row >> 2) & 3; }

    const int nchunks = d.Cin >> 5;
    const int total = nchunks * 9;
    load_norm(0);
#pragma unroll
    for (int r = 0; r < 6; ++r) store_piece(0, r, load_piece(0, r));
    issue_b(0, 0);

    float4 pend = make_float4(0.f, 0.f, 0.f, 0.f);       // halo piece in flight (issued in the previous step)
    int c = 0, t = 0;
    for (int s = 0; s < total; ++s) {
        __syncthreads();
        const bool more = s + 1 < total;
        int nc = c, nt = t + 1;
        if (nt == 9) { nt = 0; nc = c + 1; }
        const bool next_chunk = c + 1 < nchunks;
        // halo of the next chunk: piece t-1 (loaded during the previous step) is split and written, piece t is issued
        if (next_chunk) {
            if (t >= 1 && t <= 6) store_piece((c + 1) & 1, t - 1, pend);
            if (t == 0) load_norm(c + 1);
            if (t < 6) pend = load_piece(c + 1, t);
        }
        if (more) issue_b(nt * d.Cin + (nc << 5), (s + 1) & 1);
        const char* Ah = smem + (c & 1) * STAGE_A;
        const char* Bh = smem + 2 * STAGE_A + (s & 1) * STAGE_B;
        const int ty = (t * 21846) >> 16, tx = t - ty * 3;
        int a_off[TM], a_sw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int h = a_h0[i] + ty * HALO_W + tx; a_off[i] = h * ROWB; a_sw[i] = (h >> 2) & 3; }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = 2 * kk + fh;
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int o = a_off[i] + ((ch ^ a_sw[i]) << 4);
                ah[i] = *(const f16x8*)(Ah + o); al[i] = *(const f16x8*)(Ah + A_BYTES + o);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int o = b_off[j] + ((ch ^ b_sw[j]) << 4);
                bh[j] = *(const f16x8*)(Bh + o); bl[j] = *(const f16x8*)(Bh + B_BYTES + o);
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
        c = nc; t = nt;
    }

    // ---------------------------------------------------------------- epilogue: fp32 straight from the accumulators
    float* outp = (float*)d.out;
    const float* resp = (const float*)d.res;
    float* sl = (float*)smem;                 // [WGM][BN][2]
    if (d.stats) __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int lcol = wn * WTN + j * 32 + fr;
        const int col = tile_n * BN + lcol;
        const float bv = d.bias ? d.bias[col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        float rr[TM][16];                      // residual values of this column: all loads issued before any is consumed
        if (resp) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    const int y = y0 + (row >> 4), x = x0 + (row & 15);
                    const bool ok = y < d.H && x < d.W && col < d.Cout;
                    rr[i][e] = resp[ok ? (((long)n * d.H + y) * d.W + x) * d.Cout + col : 0];
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                float v = acc[i][j][e] + accl[i][j][e] * LO_INV + bv;
                s1 += v; s2 += v * v;
                const int y = y0 + (row >> 4), x = x0 + (row & 15);
                if (y >= d.H || x >= d.W || col >= d.Cout) continue;
                const long off = (((long)n * d.H + y) * d.W + x) * d.Cout + col;
                if (resp) v += rr[i][e];
                if (d.relu) v = fmaxf(v, 0.f);
                outp[off] = v;
            }
        if (d.stats) {
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (fh == 0) { sl[(wm * BN + lcol) * 2 + 0] = s1; sl[(wm * BN + lcol) * 2 + 1] = s2; }
        }
    }
    if (d.stats) {
        __syncthreads();
        if (tid < BN * 2) {                    // two 128-row records per tile: wave rows {0,1} and {2,3}
            const int rec = tid / BN, col = tid % BN;
            const float s1 = sl[((rec * 2) * BN + col) * 2 + 0] + sl[((rec * 2 + 1) * BN + col) * 2 + 0];
            const float s2 = sl[((rec * 2) * BN + col) * 2 + 1] + sl[((rec * 2 + 1) * BN + col) * 2 + 1];
            const int gcol = tile_n * BN + col;
            if (gcol < d.Cout) {
                float* dst = d.stats + ((long)(d.stats_tile_base + tile_m * 2 + rec) * 2) * d.Cout + gcol;
                dst[0] = s1; dst[d.Cout] = s2;
            }
        }
    }
}

}  // namespace

bool gdt_conv_halo_x3_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_HALO"); return e ? atoi(e) : 1; }();   // 0 off, 1 auto, 2 force
    if (mode == 0) return false;
    const bool shape = d.ntaps == 9 && d.TW == 3 && d.sy == 1 && d.sx == 1 && d.dy0 == -1 && d.dx0 == -1 && d.dys == 1 && d.dxs == 1 &&
                       d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0 && d.Cin % 32 == 0 && !d.out_f32 && d.OH == d.H &&
                       d.OW == d.W && d.Kpad == 9 * d.Cin && d.CoutPad % 128 == 0 && d.w_lo != nullptr;
    if (!shape) return false;
    if (d.stats && ((d.H & 15) || (d.W & 15))) return false;
    if (mode == 2) return true;
    const long tiles = (long)d.N * ((d.W + 15) / 16) * ((d.H + 15) / 16);
    const double useful = (double)d.H * d.W / ((double)((d.H + 15) / 16 * 16) * ((d.W + 15) / 16 * 16));
    return tiles * (d.CoutPad / 128) >= 512 && useful >= 0.85;
}

int gdt_launch_conv_halo_x3(const ConvLaunch& d, hipStream_t stream) {
    const int tiles = d.N * ((d.W + 15) / 16) * ((d.H + 15) / 16), ntn = d.CoutPad / BN;
    constexpr size_t lds = 2 * (size_t)STAGE_A + 2 * (size_t)STAGE_B;
    static_assert(lds <= 160 * 1024 && (size_t)4 * BN * 8 <= lds, "LDS budget");
    static GdtPerDevice per_dev;          // one attribute call per template instantiation AND device (gdt_common.h)
    int attr_set = 0;
    {
        const int rc = gdt_per_device(per_dev, attr_set, [](int, int, int& v) {
            v = 1;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    hipLaunchKernelGGL(conv3x3_halo_x3_kernel, dim3(gdt_grid_for_tiles(tiles, ntn)), dim3(NT), lds, stream, d);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}
