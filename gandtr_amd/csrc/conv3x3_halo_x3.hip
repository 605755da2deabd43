// 3x3 / stride-1 / pad-1 convolution in the "f16x3" precision mode (fp32 NHWC activations, split-fp16 operands, three MFMA
// passes -- see conv_igemm_x3.hip) with the LDS-resident input halo of conv3x3_halo.hip.
//
// One workgroup = 16x16 output patch x 128 output channels, 8 wavefronts (4 x 2, 64x64 per wave, two accumulator sets).
// Per 32-channel chunk the 18x18 fp32 halo is read once through registers (global_load_dwordx4), optionally normalised
// (fused InstanceNorm + ReLU of the producer: x -> max((x - mean) * rstd, 0), p2p_networks.py:29,:272; round 5: also the
// second norm of a ResnetBlock, x + IN(conv(.)) p2p_networks.py:503-506, with the transformed tensor written back for the
// block's later consumers by the column tile 0 workgroup -- interior pixels of its patch only), split into
// hi / lo fp16 images and written to LDS; the 9 taps then run against it.  The pre-split weights stream per tap with
// global_load_lds.  Halo pieces of the next chunk are issued one per tap step and written one step later
// (issue-early / write-late), so their latency hides under a whole step of MFMAs.
//
// FORM 1 / 2 (round 5): the generator's shift layers on the same skeleton instead of the generic K-step-32 GEMM (conv_igemm_x3.hip gathers every input
// pixel once per tap; 91-115 TFLOP/s):
//   FORM 1  any tap table inside the 3 x 3 window with a strided output grid -- the four sub-pixel phase launches of ConvTranspose2d(k3,s2,p1,op1)
//           (p2p_networks.py:295-300; 1 / 2 / 2 / 4 taps, output pixel (2y + py, 2x + px));
//   FORM 2  Conv2d(k3,s2,p1) (p2p_networks.py:278-280) as a 2 x 2-shift convolution over the virtual space-to-depth view of its input (conv3x3_halo_c.hip FORM 2:
//           K index = shift * 4 Cin + parity * Cin + c); a 32-channel chunk has one parity, the (shift, parity) pairs that do not occur are skipped as whole steps.
//   With 1-4 steps per chunk there is no tap step per halo piece: the six pieces of the next chunk are issued in the chunk's first step and written after the
//   MFMAs of its last one.
#include <cstdlib>

#include "gdt_common.h"

// Where a step's time goes (round 5, FORM 0 at 64 x 256^2: 0.93 ms per launch, 18 launches = 70 % of the exact mode's forward; 2 waves per SIMD, one step = one tap
// of one 32-channel chunk = 24 MFMAs per wave behind one workgroup barrier):
//   * timing-only ablations (-DGDT_X3_ABL=bits: 1 no halo loads after the first chunk, 2 no MFMAs, 4 no weight staging after the first step; 18 launches): 17.1 ms as
//     is; 15.2 / 10.8 / 15.4 with bit 1 / 2 / 4 alone, 14.0 with 1 + 4, 6.5 with all three;
//   * s_memtime stamps, shader cycles per step, waves 0-3 / 4-7: barrier wait 880 / 160, fragment reads + load issue 920 / 1830, the 24 MFMAs 780 / 860 (full rate
//     while they run), staging arithmetic 650 / 400: 3250 in all, of which the matrix pipe runs 1640.  The older wave of a SIMD issues first; the younger one's reads and
//     MFMAs queue behind it; nothing of the front part or the staging lies under an MFMA;
//   * SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.32 (the 64-byte halo rows; conv3x3_halo_c16.hip's pixel <-> column map has 0.02), LDS active 21 % of the CU's cycles;
//   * tried, all within +-3 % or slower (each passed the parity tests): both 16-k halves' fragment reads issued ahead of the MFMAs; the two waves of a SIMD running
//     MFMAs / staging in opposite order; fragments pipelined ACROSS the step barrier with a third weight stage (the next step's first set read under this step's
//     second MFMA half), as written and with a branch-free front part (every address made valid by mask arithmetic, so that the front is one basic block: the
//     compiler still puts an lgkmcnt(0) in front of the first MFMA of the loop body, and the unconditional loads cost 1.1 ms); and a loop whose steps BEGIN with
//     their MFMAs (first fragment set read at the end of the previous step, second set behind the first MFMA half; the ISA shows 12 MFMAs, 8 reads, the global issue,
//     12 MFMAs, 8 reads, staging, barrier -- exactly as meant): 2885 images/s against 2952 for this plain loop, alternating;
//   * why none of it moves: the forward runs AT THE POWER CAP.  tools/x3_clock.py, 3 s of exact-mode forwards: 1376 W of the 1400 W cap, 2.07 GHz (plain loop) /
//     2.10 GHz (MFMA-first loop, and slower); the default mode's forward 1361 W at 1.96 GHz.  A better schedule is handed back as clock; what counts is energy per
//     result -- fewer instructions and bytes (the 3-VALU split below was worth 1-2 %), not fewer stalls.  The kernel stays at 0.42 of the mode's three-pass ceiling:
//     331 TFLOP/s algorithmic = 1.0 PFLOP/s of MFMA work, the rate the compensated resblock kernel also runs at.
#ifndef GDT_X3_ABL
#define GDT_X3_ABL 0
#endif

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

template <int FORM>
__global__ __launch_bounds__(NT) void conv3x3_halo_x3_kernel(const ConvLaunch d) {
    constexpr bool BURST = FORM != 0, S2D = FORM == 2;
    constexpr int WGN = 2, WTM = 64, WTN = 64, TM = 2, TN = 2;      // 4 x 2 wavefronts, 64 x 64 per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2][A hi, A lo] [2][B hi, B lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const float* __restrict__ in = (const float*)d.in;

    // the patch grid: the output grid of the launch (FORM 0: = input = output; 1: the phase grid = input; 2: the output = space-to-depth grid)
    const int GH = d.OHg, GW = d.OWg;
    const int lcr = d.lc8 + 3 - (S2D ? 2 : 0);                       // log2 of the REAL input channel count (FORM 2: d.Cin counts the 4 parities)
    const int tiles_x = (GW + 15) >> 4, tiles_y = (GH + 15) >> 4;
    const int tpi = tiles_x * tiles_y, ntm = d.N * tpi, ntn = d.CoutPad / BN;
    int tile_m, tile_n;
    if (!gdt_tile_of_block(blockIdx.x, ntm, ntn, tile_m, tile_n)) return;      // XCD-chunked, see gdt_common.h
    const int n = tile_m / tpi, tr = tile_m - n * tpi;
    const int y0 = (tr / tiles_x) << 4, x0 = (tr % tiles_x) << 4;

    // ---- halo staging state: this thread owns the 4-channel group c4 = tid & 7 of halo rows (tid >> 3) + 64*r
    const int c4 = tid & 7, hrow = tid >> 3;
    const bool refl = d.pad_reflect != 0;
    const float* __restrict__ inres = (const float*)d.in_res;
    float* __restrict__ inout = tile_n == 0 ? (float*)d.in_out : nullptr;
    int a_pix[6]; unsigned a_ok = 0, a_int = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int h = r * 64 + hrow;
        const int hy = h / HALO_W, hx = h - hy * HALO_W;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const int IH = S2D ? GH : d.H, IW = S2D ? GW : d.W;        // (FORM 2: halo coordinates are space-to-depth pixels (R, C) = input pixels (2R + py, 2C + px))
        int ry = iy < 0 ? -iy : (iy >= IH ? 2 * IH - 2 - iy : iy);
        int rx = ix < 0 ? -ix : (ix >= IW ? 2 * IW - 2 - ix : ix);
        ry = min(max(ry, 0), IH - 1); rx = min(max(rx, 0), IW - 1);
        const bool inb = ((unsigned)iy < (unsigned)IH) & ((unsigned)ix < (unsigned)IW);
        a_pix[r] = S2D ? (n * d.H + 2 * ry) * d.W + 2 * rx : (n * d.H + ry) * d.W + rx;
        a_ok |= ((h < HALO_ROWS) & (inb | refl) ? 1u : 0u) << r;
        a_int |= (((h < HALO_ROWS) & inb & (hy >= 1) & (hy <= 16) & (hx >= 1) & (hx <= 16)) ? 1u : 0u) << r;      // the patch's own pixels (write-back)
    }
    float4 nm0 = make_float4(0.f, 1.f, 0.f, 1.f), nm1 = nm0;        // (mean, rstd) x 4 channels of the chunk being staged
    // channel offset and pixel offset of a chunk: FORM 2 -- parity (chunk * 32) / Cin of the space-to-depth view, real channels (chunk * 32) % Cin
    auto chan_of = [&](int chunk) { return S2D ? ((chunk << 5) & ((1 << lcr) - 1)) + c4 * 4 : (chunk << 5) + c4 * 4; };
    auto pix_of = [&](int chunk) { const int par = (chunk << 5) >> lcr; return S2D ? (par >> 1) * d.W + (par & 1) : 0; };
    auto load_norm = [&](int chunk) {
        if (!d.in_norm) return;
        const float* p = d.in_norm + (((long)n << lcr) + chan_of(chunk)) * 2;
        nm0 = *(const float4*)p; nm1 = *(const float4*)(p + 4);
    };
    // a piece = the raw conv output (+ the raw residual); normalisation, residual add, write-back and the split happen one tap step later, at the LDS write
    // (write-late: nothing waits for a load in the step that issued it)
    struct Piece { float4 v, rv; };
    auto load_piece = [&](int chunk, int r) -> Piece {
        Piece p;
        p.v = p.rv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r * 64 + hrow >= HALO_ROWS_PAD) return p;
        if (!((a_ok >> r) & 1u)) return p;
        if ((GDT_X3_ABL & 1) && chunk > 0) return p;
        const long off = ((long)(a_pix[r] + pix_of(chunk)) << lcr) + chan_of(chunk);
        p.v = *(const float4*)(in + off);
        if (!BURST && inres) p.rv = *(const float4*)(inres + off);
        return p;
    };
    auto store_piece = [&](int stage, int chunk, int r, const Piece& p) {
        const int row = r * 64 + hrow;
        if (row >= HALO_ROWS_PAD) return;
        float4 v = p.v;
        if ((a_ok >> r) & 1u) {                  // (padding pieces stay zero)
            if (d.in_norm) {
                v.x = (v.x - nm0.x) * nm0.y; v.y = (v.y - nm0.z) * nm0.w; v.z = (v.z - nm1.x) * nm1.y; v.w = (v.w - nm1.z) * nm1.w;
                if (d.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            if (!BURST && inres) { v.x += p.rv.x; v.y += p.rv.y; v.z += p.rv.z; v.w += p.rv.w; }
            if (!BURST && inout && ((a_int >> r) & 1u)) *(float4*)(inout + (((long)a_pix[r] << lcr) + chunk * 32 + c4 * 4)) = v;
        }
        char* Ah = smem + stage * STAGE_A;
        const int off = row * ROWB + (((c4 >> 1) ^ ((row >> 2) & 3)) << 4) + (c4 & 1) * 8;
        // hi = fp16(x) two per instruction, x - hi in one v_fma_mix each (the same values as (f16)x, (f16)((x - (float)hi) * 2^11): 3 VALU per element instead of 5)
        const float x[4] = {v.x, v.y, v.z, v.w};
        unsigned hw[2], lw[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float l0, l1;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hw[q]) : "v"(x[2 * q]), "v"(x[2 * q + 1]));
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hw[q]), "v"(x[2 * q]));
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hw[q]), "v"(x[2 * q + 1]));
            l0 *= LO_SCALE; l1 *= LO_SCALE;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lw[q]) : "v"(l0), "v"(l1));
        }
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        *(u32x2*)(Ah + off) = u32x2{hw[0], hw[1]};
        *(u32x2*)(Ah + A_BYTES + off) = u32x2{lw[0], lw[1]};
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

    const int nchunks = d.Cin >> 5, ntaps = d.ntaps;
    // the steps: (chunk, tap) pairs in order.  FORM 2 skips the (shift, parity) pairs that do not occur in a stride-2 3x3 conv: shift -1 meets parity 1 only
    // (kernel row 0), shift 0 both (rows 1, 2) -- rows and columns alike; every chunk keeps the (0, 0) shift
    auto live = [&](int cc, int tt) -> bool {
        if (!S2D) return true;
        const int par = (cc << 5) >> lcr;
        return (((tt >> 1) | (par >> 1)) & ((tt & 1) | (par & 1))) != 0;
    };
    auto advance = [&](int& cc, int& tt) { do { if (++tt == ntaps) { tt = 0; ++cc; } } while (cc < nchunks && !live(cc, tt)); };
    int c = 0, t = 0;
    if (!live(0, 0)) advance(c, t);
    load_norm(0);
#pragma unroll
    for (int r = 0; r < 6; ++r) store_piece(0, 0, r, load_piece(0, r));
    issue_b(t * d.Cin, 0);

    Piece pend[BURST ? 6 : 1];                           // halo pieces in flight (FORM 0: the one issued in the previous step)
#pragma unroll
    for (int r = 0; r < (BURST ? 6 : 1); ++r) pend[r].v = pend[r].rv = make_float4(0.f, 0.f, 0.f, 0.f);
    bool first_of_chunk = true;
    for (int s = 0; c < nchunks; ++s) {
        __syncthreads();
        int nc = c, nt = t;
        advance(nc, nt);
        const bool more = nc < nchunks, last_of_chunk = nc != c;
        const bool next_chunk = c + 1 < nchunks;
        if (next_chunk) {
            if (!BURST) {
                // halo of the next chunk: piece t-1 (loaded during the previous step) is split and written, piece t is issued
                if (t >= 1 && t <= 6) store_piece((c + 1) & 1, c + 1, t - 1, pend[0]);
                if (t == 0) load_norm(c + 1);
                if (t < 6) pend[0] = load_piece(c + 1, t);
            } else if (first_of_chunk) {
                load_norm(c + 1);                            // (this chunk's pieces were written at the end of the previous one)
#pragma unroll
                for (int r = 0; r < 6; ++r) pend[BURST ? r : 0] = load_piece(c + 1, r);
            }
        }
        if (more && !(GDT_X3_ABL & 4)) issue_b(nt * d.Cin + (nc << 5), (s + 1) & 1);
        const char* Ah = smem + (c & 1) * STAGE_A;
        const char* Bh = smem + 2 * STAGE_A + (s & 1) * STAGE_B;
        const int ty = (t * d.invTW) >> 16, tx = t - ty * d.TW;
        const int tap_h = (d.dy0 + ty * d.dys + 1) * HALO_W + d.dx0 + tx * d.dxs + 1;       // the tap's offset in the halo (taps lie inside the 3 x 3 window)
        int a_off[TM], a_sw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int h = a_h0[i] + tap_h; a_off[i] = h * ROWB; a_sw[i] = (h >> 2) & 3; }
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
            for (int i = 0; i < ((GDT_X3_ABL & 2) ? 0 : TM); ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accl[i][j], 0, 0, 0);
                    accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accl[i][j], 0, 0, 0);
                }
        }
        if (BURST && next_chunk && last_of_chunk) {
            __builtin_amdgcn_sched_barrier(0);               // (behind the step's MFMAs: the pieces have had the chunk to arrive)
#pragma unroll
            for (int r = 0; r < 6; ++r) store_piece((c + 1) & 1, c + 1, r, pend[BURST ? r : 0]);
        }
        first_of_chunk = last_of_chunk;
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
        // (paired phases of a transposed conv: the wave's 64 columns are one phase -- its own output pixel offset, channels 0-63)
        const bool half = d.pair_cout > 0 && lcol >= d.pair_cout;
        const int co = half ? col - d.pair_cout : col;
        const int ooy = half ? d.ooy2 : d.ooy, oox = half ? d.oox2 : d.oox;
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
                    const bool ok = y < GH && x < GW && co < d.Cout;
                    rr[i][e] = resp[ok ? (((long)n * d.OH + y * d.osy + ooy) * d.OW + x * d.osx + oox) * d.Cout + co : 0];
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
                if (y >= GH || x >= GW || co >= d.Cout) continue;
                const long off = (((long)n * d.OH + y * d.osy + ooy) * d.OW + x * d.osx + oox) * d.Cout + co;
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
            const bool half = d.pair_cout > 0 && col >= d.pair_cout;            // (paired phases: the second one's records lie one set = M / 128 records further on)
            const int gco = half ? gcol - d.pair_cout : gcol;
            if (gco < d.Cout) {
                float* dst = d.stats + ((long)(d.stats_tile_base + (half ? d.M / 128 : 0) + tile_m * 2 + rec) * 2) * d.Cout + gco;
                dst[0] = s1; dst[d.Cout] = s2;
            }
        }
    }
}

}  // namespace

bool gdt_conv_halo_x3_eligible(const ConvLaunch& d) {
    const char* e = getenv("GDT_CONV_HALO_X3");                 // 0 off, 1 auto, 2 force (read per call: tests force the patch kernels at small batches)
    const int mode = e ? atoi(e) : 1;
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

// FORM 1 (a tap table inside the 3 x 3 window, strided output: the phase launches of a transposed conv) and FORM 2 (d.x3_form == 2: stride-2 3x3 conv over the
// virtual space-to-depth view, net.hip s2_geometry); plain InstanceNorm (+ReLU) folding only
bool gdt_conv_halo_x3_taps_eligible(const ConvLaunch& d) {
    const char* e = getenv("GDT_CONV_HALO_X3_FORMS");           // 0 off (A/B: the generic GEMM), 1 auto, 2 force (read per call)
    const int mode = e ? atoi(e) : 1;
    if (mode == 0 || !d.w_lo || d.out_f32 || d.in_res || d.in_out || d.pool2 || d.CoutPad % 128 != 0 || d.osy < 1 || d.osx < 1) return false;
    if (d.pair_cout && (d.pair_cout != 64 || d.CoutPad != 128 || d.Cout != 64 || d.x3_form != 0)) return false;      // paired phases: one 128-column tile = 2 x 64 channels
    if (d.x3_form == 2) {
        const bool shape = d.ntaps == 4 && d.TW == 2 && d.dy0 == -1 && d.dys == 1 && d.dx0 == -1 && d.dxs == 1 && d.sy == 2 && d.sx == 2 && d.osy == 1 && d.osx == 1 &&
                           d.ooy == 0 && d.oox == 0 && !d.pad_reflect && !(d.H & 1) && !(d.W & 1) && d.OH == d.H / 2 && d.OW == d.W / 2 && d.OHg == d.OH && d.OWg == d.OW &&
                           d.Cin % 128 == 0 && d.Kpad == 4 * d.Cin;
        if (!shape) return false;
    } else {
        if (d.x3_form != 0 || d.sy != 1 || d.sx != 1 || d.ntaps < 1 || d.ntaps > 9 || d.TW < 1 || d.Cin % 32 != 0 || d.Kpad != d.ntaps * d.Cin || d.OHg != d.H || d.OWg != d.W) return false;
        for (int t = 0; t < d.ntaps; ++t) {
            const int dy = d.dy0 + (t / d.TW) * d.dys, dx = d.dx0 + (t % d.TW) * d.dxs;
            if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
        }
    }
    if ((long)d.N * d.H * d.W * (d.x3_form == 2 ? d.Cin / 4 : d.Cin) >= (1L << 31)) return false;
    if (d.stats && ((d.OHg & 15) || (d.OWg & 15))) return false;
    if (mode == 2) return true;
    const long tiles = (long)d.N * ((d.OWg + 15) / 16) * ((d.OHg + 15) / 16);
    const double useful = (double)d.OHg * d.OWg / ((double)((d.OHg + 15) / 16 * 16) * ((d.OWg + 15) / 16 * 16));
    return tiles * (d.CoutPad / 128) >= 512 && useful >= 0.85;
}

template <int FORM>
static int launch_halo_x3(const ConvLaunch& d, hipStream_t stream) {
    const int tiles = d.N * ((d.OWg + 15) / 16) * ((d.OHg + 15) / 16), ntn = d.CoutPad / BN;
    constexpr size_t lds = 2 * (size_t)STAGE_A + 2 * (size_t)STAGE_B;
    static_assert(lds <= 160 * 1024 && (size_t)4 * BN * 8 <= lds, "LDS budget");
    static GdtPerDevice per_dev;          // one attribute call per template instantiation AND device (gdt_common.h)
    int attr_set = 0;
    {
        const int rc = gdt_per_device(per_dev, attr_set, [](int, int, int& v) {
            v = 1;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_x3_kernel<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    hipLaunchKernelGGL(conv3x3_halo_x3_kernel<FORM>, dim3(gdt_grid_for_tiles(tiles, ntn)), dim3(NT), lds, stream, d);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_launch_conv_halo_x3(const ConvLaunch& d, hipStream_t stream) { return launch_halo_x3<0>(d, stream); }
int gdt_launch_conv_halo_x3_taps(const ConvLaunch& d, hipStream_t stream) { return d.x3_form == 2 ? launch_halo_x3<2>(d, stream) : launch_halo_x3<1>(d, stream); }
