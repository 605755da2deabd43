// fp16 NHWC epilogue shared by conv_igemm.hip and conv3x3_halo.hip.
//
// Input: the wave's 32x32 MFMA accumulator tiles acc[TM][TN] (C layout of v_mfma_f32_32x32x16: column = lane & 31, row =
// (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).  Steps: + bias (folded BatchNorm shift), optional ReLU, fp16 -> LDS transpose
// so that every lane stores 16 contiguous bytes; per-128-row InstanceNorm partial statistics (sum, sum of squares) from the
// fp32 values with a fixed reduction order (deterministic); optional residual add (+ ReLU) in fp32 on the way out.
// `pixel_of(row, ok)` maps a tile row to the output pixel index (n * OH * OW + y * OW + x) and says whether it exists.
#pragma once
#include "gdt_common.h"

template <int BM, int BN, int WGM, int WGN, int NT>
constexpr size_t conv_epilogue_lds_bytes() {
    return ((size_t)BM * (BN + 8) * 2 + 255) / 256 * 256 + (size_t)WGM * BN * 8;
}

template <int BM, int BN, int WGM, int WGN, int NT, int TM, int TN, typename PixelOf>
__device__ __forceinline__ void conv_epilogue_f16(const ConvLaunch& d, const f32x16 (&acc)[TM][TN], char* smem, int tile_m, int tile_n,
                                                  PixelOf pixel_of) {
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int CP = BN + 8;                                      // padded C-tile row (halves)
    constexpr int STATS_OFF = (BM * CP * 2 + 255) / 256 * 256;      // [WGM][BN][2] floats behind the C tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN, fr = lane & 31, fh = lane >> 5;
    __syncthreads();                                                // all MFMA reads of the staging buffers are done
    f16* Ct = (f16*)smem;
    float* sl = (float*)(smem + STATS_OFF);
    const bool relu_now = d.relu && !d.res;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * WTN + j * 32 + fr;
        const float bv = d.bias ? d.bias[tile_n * BN + col] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                float v = acc[i][j][e] + bv;
                s1 += v; s2 += v * v;
                if (relu_now) v = fmaxf(v, 0.f);
                Ct[row * CP + col] = (f16)v;
            }
        if (d.stats) {      // per-lane column sums over the wave's rows, the two half-waves combined
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (fh == 0) { sl[(wm * BN + col) * 2 + 0] = s1; sl[(wm * BN + col) * 2 + 1] = s2; }
        }
    }
    __syncthreads();
    constexpr int RT = BM / 128;            // 128-row statistics records per tile
    constexpr int WPR = WGM / RT;           // wave rows per record
    static_assert(BM % 128 == 0 && WGM % RT == 0 && BN * RT <= NT, "statistics record layout");
    if (d.stats && tid < BN * RT) {
        const int rec = tid / BN, col = tid % BN;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WPR; ++w) { s1 += sl[((rec * WPR + w) * BN + col) * 2 + 0]; s2 += sl[((rec * WPR + w) * BN + col) * 2 + 1]; }
        const int gcol = tile_n * BN + col;
        // (a 256-row tile whose second half lies past M -- M % 256 == 128 -- has no record slot: the slab holds M / 128 records)
        if (gcol < d.Cout && tile_m * RT + rec < d.M / 128) {
            float* dst = d.stats + ((long)(d.stats_tile_base + tile_m * RT + rec) * 2) * d.Cout + gcol;
            dst[0] = s1; dst[d.Cout] = s2;
        }
    }
    if (d.dbg & 8) return;                  // timing-only ablation: no global stores
    constexpr int CPR = BN / 8;             // 16-byte chunks per tile row
    if (d.pool2) {
        // fused MaxPool2d(2, 2) (patch kernels only: tile row = py * 16 + px): one pooled pixel = max over rows r, r+1, r+16, r+17
        for (int id = tid; id < (BM / 4) * CPR; id += NT) {
            const int chunk = id % CPR, pp = id / CPR;
            const int pr = pp >> 3, pc = pp & 7, row0 = pr * 32 + 2 * pc;
            const f16* r0 = Ct + row0 * CP + chunk * 8;
            f16x8 v = __builtin_elementwise_max(*(const f16x8*)r0, *(const f16x8*)(r0 + CP));
            v = __builtin_elementwise_max(v, __builtin_elementwise_max(*(const f16x8*)(r0 + 16 * CP), *(const f16x8*)(r0 + 17 * CP)));
            bool ok;
            const long pix = pixel_of(row0, ok);                      // (n * H + y) * W + x of the window's top-left pixel
            const int col = tile_n * BN + chunk * 8;
            if (ok && col < d.Cout) {
                const long q = pix / d.W; const int x = (int)(pix - q * d.W);
                const long n = q / d.H; const int y = (int)(q - n * d.H);
                if ((y + 1 < d.H) & (x + 1 < d.W))          // (odd H / W: the last row / column is dropped, as MaxPool2d(2, 2) does)
                    *(f16x8*)(d.out + (((n * (d.H >> 1) + (y >> 1)) * (d.W >> 1) + (x >> 1)) * d.Cout + col)) = v;
            }
        }
        return;
    }
    constexpr int NCH = BM * CPR / NT;      // chunks per thread
    // All residual loads are issued first (the accumulators are dead by now, so registers are plentiful) and consumed
    // afterwards.  With the load inside the per-chunk branch each lane had ONE residual load in flight and the read ran at
    // 2.8 TB/s (the ResNet-101 expansion convs spent 40 % of their time there: 0.248 -> 0.204 ms).  Issuing them before the
    // LDS transpose instead spills in the 256x256 tiles and measured slower (0.242 ms).
    long offs[NCH];
    f16x8 rv[NCH];
    unsigned okmask = 0;
    const bool has_res = d.res && !(d.dbg & 32);      // (dbg 32: timing-only ablation without the residual read)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int id = c * NT + tid;
        const int col = tile_n * BN + (id % CPR) * 8;
        bool ok;
        const long pix = pixel_of(id / CPR, ok);
        ok = ok && col < d.Cout;
        offs[c] = ok ? pix * d.Cout + col : 0;
        okmask |= (ok ? 1u : 0u) << c;
        if (has_res) rv[c] = *(const f16x8*)(d.res + offs[c]);          // offset 0 is a valid address for masked chunks
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int id = c * NT + tid;
        f16x8 v = *(const f16x8*)(Ct + (id / CPR) * CP + (id % CPR) * 8);
        if (has_res) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = (float)v[e] + (float)rv[c][e];
                if (d.relu) t = fmaxf(t, 0.f);
                v[e] = (f16)t;
            }
        }
        if ((okmask >> c) & 1u) *(f16x8*)(d.out + offs[c]) = v;
    }
}
