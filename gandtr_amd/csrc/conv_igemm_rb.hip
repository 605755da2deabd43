// Persistent implicit-GEMM convolution with the weights streamed into registers (gfx950, MI355X).
//
// Second generation of conv_igemm.hip for the layers that are NOT 3x3 / stride 1 (those have conv3x3_halo_rb.hip): the 1x1
// convs of the ResNet-101 Bottlenecks (torchvision resnet.py Bottleneck.conv1 / conv3 / downsample, sliced at
// imageretrievalnet.py:185-190), the stride-2 3x3 down-sampling convs and the ConvTranspose2d phases of the generator
// (p2p_networks.py:282-311).  GEMM view as conv_igemm.hip: M = N*OHg*OWg, K = taps*Cin (k = tap*Cin + c), Cin % 64 == 0.
//
// What the first kernel measured (ResNet-101, batch 32 @ 1024^2, 62 launches of its 256x256 tile = 11.9 ms): skipping the
// output stores cut the time to 6.5 ms.  With a handful of K-steps per tile (K = 256 .. 1024), every CU reaches its epilogue
// at the same moment, so a layer alternates between a phase that only computes and a phase that only moves output /
// residual bytes -- neither MFMA nor HBM is busy half of the time.  Here:
//   * PERSISTENT grid (one workgroup per CU walking the XCD-chunked tile list): the epilogue's global stores drain while the
//     next tile's K-loop runs, and the first A tile / weight slice of the next tile are fetched during the last steps of
//     the current one;
//   * weights straight from L2 into the MFMA B registers in fragment order (conv3x3_halo_rb.hip): no weight stages in LDS,
//     half the LDS traffic, room for an epilogue region of its own;
//   * the A tile (256 rows x 64 channels per K-step) goes through registers, two steps ahead (load in step g, LDS write in
//     step g+1, used in step g+2), with the tap shift / padding resolved per 16-byte piece; the producer's InstanceNorm
//     (+ReLU) can be applied on the way (v_fma_mix, see conv3x3_halo_rb.hip);
//   * everything in the K-loop is branch-free with respect to memory operations, so the compiler's s_waitcnt are counted.
#include <cstdio>
#include <cstdlib>

#include "gdt_common.h"

namespace {

constexpr int ROWB = 128;          // bytes per LDS row (64 halves of K)
constexpr int BM = 256;
constexpr int A_BYTES = BM * ROWB;                         // 32 KB per stage
constexpr int NORM_BYTES = 4096 + 64;                      // (scale, shift): two slots of up to 256 input channels + a zero entry
constexpr int C_OFF = 2 * A_BYTES + NORM_BYTES;            // epilogue transpose region (half a C tile)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned norm_pair(unsigned raw, float s0, float h0, float s1, float h1) {
    unsigned o;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(o) : "v"(raw), "v"(s0), "v"(h0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(o) : "v"(raw), "v"(s1), "v"(h1));
    return o;
}

// epilogue region: half a C tile (each wave's row blocks in two halves), or the whole tile when a wave owns a single row block
template <int BN, int WGM>
constexpr int irb_region_rows() { return (BM / WGM) / 32 >= 2 ? BM / 2 : BM; }
template <int BN, int WGM>
constexpr size_t irb_lds_bytes() { return (size_t)C_OFF + (size_t)8 * 32 * (64 + 8) * 2; }        // + eight wave-private transpose patches

struct TileAt { int tile_m, tile_n; bool valid; };

template <int BN, int WGM, int WGN, bool NORM, bool CTF = false>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_igemm_rb_kernel(const ConvLaunch d, const int vblocks) {
    constexpr int NT = WGM * WGN * 64, RPR = NT / 8, AR = BM / RPR;      // 512 threads, 64 rows per loader round, 4 rounds
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert(NT == 512 && TM >= 1 && TN >= 1 && (TM % 2 == 0 || WGM == 8), "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 31, fh = lane >> 5;

    const int ntm = (d.M + BM - 1) / BM, ntn = d.CoutPad / BN;
    auto tile_at = [&](int vb) -> TileAt {
        TileAt t;
        t.valid = vb < vblocks && gdt_tile_of_block(vb, ntm, ntn, t.tile_m, t.tile_n);
        if (!t.valid) { t.tile_m = 0; t.tile_n = 0; }
        return t;
    };
    int vb = blockIdx.x;
    TileAt cur = tile_at(vb);
    if (!cur.valid) return;                   // (validity is monotone in vb)

    const int hw_g = d.OHg * d.OWg;
    const int nk = d.Kpad >> 6, cpt = d.Cin >> 6, nks = d.Kpad >> 4;     // K-steps per tile, 64-channel chunks per tap
    const bool refl = d.pad_reflect != 0;

    // ---- A staging cursor: (tile, K-step) of the next piece set to LOAD, two compute steps ahead
    const int lrow = tid >> 3;                                 // 0..63; this thread's rows are lrow + 64 r
    const int q = (lane & 7) ^ ((lrow >> 1) & 7);              // source chunk of its 16-byte piece (XOR swizzle; 64 r keeps it)
    int s_pix0[AR], s_iy0[AR], s_ix0[AR]; unsigned s_valid = 0;
    int s_n = 0;                                               // image of the staged tile (NORM: whole tiles lie in one image)
    auto seat = [&](const TileAt& ta) {                        // per-row output position of the tile being staged
        s_valid = 0;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int m = ta.tile_m * BM + r * RPR + lrow;
            const int mm = m < d.M ? m : 0;
            const int n = mm / hw_g, rem = mm - n * hw_g;
            const int oy = rem / d.OWg, ox = rem - oy * d.OWg;
            s_pix0[r] = n * d.H * d.W; s_iy0[r] = oy * d.sy; s_ix0[r] = ox * d.sx;
            s_valid |= (m < d.M ? 1u : 0u) << r;
        }
        s_n = (ta.tile_m * BM) / hw_g;
    };
    int s_tap = 0, s_chunk = 0, s_step = 0;                    // cursor within the staged tile
    int s_slot = 0;                                            // NORM: (scale, shift) slot of the staged tile
    struct Pend { f16x8 v[AR]; unsigned ok; int slot; int chunk; };
    auto load_pend = [&]() -> Pend {
        Pend p; p.ok = 0; p.slot = s_slot; p.chunk = s_chunk;
        const int ty = (s_tap * d.invTW) >> 16, tx = s_tap - ty * d.TW;
        const int dy = d.dy0 + ty * d.dys, dx = d.dx0 + tx * d.dxs;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int iy = s_iy0[r] + dy, ix = s_ix0[r] + dx;
            int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
            int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
            ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);           // always a valid address
            const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
            const bool ok = (((s_valid >> r) & 1u) != 0) & (inb | refl);
            p.ok |= (ok ? 1u : 0u) << r;
            const unsigned off = ((unsigned)(s_pix0[r] + ry * d.W + rx) << (d.lc8 + 3)) + (s_chunk * 8 + q) * 8;
            p.v[r] = *(const f16x8*)(d.in + off);
        }
        return p;
    };
    float* nlds = (float*)(smem + 2 * A_BYTES);
    constexpr int ZERO_ENTRY = 2 * 512;                        // floats
    auto stage_norm = [&](int n, int slot) {
        for (int i = tid; i < d.Cin / 2; i += NT) {            // float4 = 2 channels x (mean, rstd)
            const float4 v = *(const float4*)(d.in_norm + (long)n * d.Cin * 2 + i * 4);
            *(float4*)(nlds + slot * 512 + i * 4) = make_float4(v.y, -v.x * v.y, v.w, -v.z * v.w);
        }
        if (tid < 4) *(float4*)(nlds + ZERO_ENTRY + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_pend = [&](const Pend& p, int stage_off) {
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const bool ok = (p.ok >> r) & 1u;
            f16x8 o;
            if (!NORM) {
                f16x8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (f16)0.f;
                o = ok ? p.v[r] : z;
            } else {                                           // padded positions read the all-zero table entry
                const float4* np4 = (const float4*)(nlds + (ok ? p.slot * 512 + (p.chunk * 8 + q) * 16 : ZERO_ENTRY));
                const u32x4 rawu = __builtin_bit_cast(u32x4, p.v[r]);
                u32x4 ou;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float4 v = np4[k]; ou[k] = norm_pair(rawu[k], v.x, v.y, v.z, v.w); }
                o = __builtin_bit_cast(f16x8, ou);
                f16x8 lo8;
#pragma unroll
                for (int e = 0; e < 8; ++e) lo8[e] = d.in_relu ? (f16)0.f : (f16)-65504.f;
                o = __builtin_elementwise_max(o, lo8);
            }
            *(f16x8*)(smem + stage_off + (r * RPR + lrow) * ROWB + ((lane & 7) << 4)) = o;
        }
    };
    // advance the cursor by one K-step; past the end of a tile it moves to the next tile of this workgroup (or, when there is
    // none, parks on the current tile's first step: harmless re-reads, never consumed)
    TileAt s_tile = cur; int s_vb = vb;
    auto advance = [&]() {
        ++s_step; ++s_chunk;
        if (s_chunk == cpt) { s_chunk = 0; ++s_tap; }
        if (s_step == nk) {
            s_step = 0; s_chunk = 0; s_tap = 0;
            const TileAt nx = tile_at(s_vb + gridDim.x);
            if (nx.valid) {
                s_tile = nx; s_vb += gridDim.x;
                seat(s_tile);
                if (NORM) { s_slot ^= 1; stage_norm(s_n, s_slot); }    // published by the barrier of this compute step
            }
        }
    };

    // ---- weights: B fragments straight from the fragment-ordered copy (uniform base + lane * 16 bytes)
    const unsigned lane_off = lane * 8;
    f16x8 b[4][TN];
    auto load_b = [&](int kk, int tile_n, int step) {
        const f16* wb = d.w_frag + ((long)((tile_n * BN + wn * WTN) / 32) * nks + step * 4) * 512;      // uniform
#pragma unroll
        for (int j = 0; j < TN; ++j) b[kk][j] = *(const f16x8*)(wb + ((long)j * nks * 512 + kk * 512) + lane_off);
    };

    // A fragment address: row = wm * WTM + i * 32 + fr; the swizzle term depends on the row modulo 16 only
    const int a_lane = (wm * WTM + fr) * ROWB + ((fh ^ (((wm * WTM + fr) >> 1) & 7)) << 4);
    auto a_frag = [&](int stage_off, int i, int kk) -> f16x8 {
        return *(const f16x8*)(smem + ((a_lane + stage_off) ^ (kk << 5)) + i * 32 * ROWB);
    };

    // ---- prologue: step 0 staged synchronously, step 1 in flight, weights of step 0
    seat(cur);
    if (NORM) { stage_norm(s_n, 0); __syncthreads(); }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) load_b(kk, cur.tile_n, 0);
    {
        const Pend p0 = load_pend();
        store_pend(p0, 0);
        advance();
    }
    Pend pend = load_pend();
    advance();
    __syncthreads();

    f16x8 afr[2][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) afr[0][i] = a_frag(0, i, 0);

    int so = 0;                                   // LDS offset of the current step's A stage
    for (;;) {
        const TileAt nxt = tile_at(vb + gridDim.x);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        int c_tap = 0, c_chunk = 0;                   // (CTF) input shift of the current K-step
        for (int s = 0; s < nk; ++s) {
            const bool last = s + 1 == nk;
            // fused transposed conv: the (shift, phase) weight block of column block j is all zero unless the phase uses the
            // shift (phase 0: shift 0; 1: dx; 2: dy; 3: all) -- those MFMAs are skipped (wave-uniform, no memory operation in
            // the branch, so the counted waits stay exact).  Phases of this wave's two column blocks: gdt_ctf_column().
            bool use_j[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) use_j[j] = true;
            if (CTF) {
                const int pair = ((cur.tile_n * WGN + wn) / (d.phase_cout >> 5)) & 1;
                const unsigned m0 = pair == 0 ? 0x1u : 0x3u, m1 = pair == 0 ? 0xFu : 0x5u;     // shift masks of phases (0 | 1), (3 | 2)
                use_j[0] = (m0 >> c_tap) & 1u;
                if (TN > 1) use_j[TN - 1] = (m1 >> c_tap) & 1u;
                if (++c_chunk == cpt) { c_chunk = 0; ++c_tap; }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cu = kk & 1, nx = cu ^ 1;
                if (kk < 3) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) afr[nx][i] = a_frag(so, i, kk + 1);
                }
                if (kk == 1) store_pend(pend, A_BYTES - so);             // A of the next step (loaded one step ago)
                if (kk == 2) { if (!(d.dbg & 2)) pend = load_pend(); advance(); }          // A of the step after it
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (!CTF || use_j[j]) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[kk][j], afr[cu][i], acc[i][j], 0, 0, 0);   // D[cout][pixel]
                    }
                if (!last && !(d.dbg & 1)) load_b(kk, cur.tile_n, s + 1);       // (the next TILE's first slice is fetched after the epilogue:
                __builtin_amdgcn_sched_barrier(0);              //  32 registers the epilogue needs)
            }
            // the other stage becomes current: own LDS writes done, workgroup barrier, first fragments of the next step
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            so = A_BYTES - so;
            if (!last) {
#pragma unroll
                for (int i = 0; i < TM; ++i) afr[0][i] = a_frag(so, i, 0);
            }
        }

        // ------------------------------------------------------------ epilogue: bias, ReLU, fp16, LDS transpose in two
        // halves through a region of its own, InstanceNorm statistics records, residual (+ReLU), 16-byte stores
        if (d.dbg & 4) {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) sacc += acc[i][j][0] + acc[i][j][15];
            if (sacc == 12345.678f) d.out[0] = (f16)sacc;
        } else {
            // WAVE-PRIVATE epilogue (as conv3x3_halo_rb.hip): the MFMAs run with the operands swapped (D = W * A^T), so lane (fr, fh)
            // holds pixel fr of row block i and, in registers 4g .. 4g+3, the four consecutive output channels 8g + 4fh .. +3 of
            // column block j.  Per 32-row block the wave transposes its 32 x 64 slice through its own 4.6 KB LDS patch (8-byte
            // writes, 16-byte reads) and stores one 128-byte line per pixel; bias, ReLU, residual (+ReLU), sub-pixel scatter
            // (fused transposed conv) on the way.  No workgroup barrier.  InstanceNorm statistics: sums of the stored fp16 values
            // over the wave's WTM rows, lanes sharing a channel group merged by a fixed butterfly, written as the wave's own
            // record -- records of WTM < 128 rows go to separate record sets (gdt_conv_igemm_rb_stats_sets) the finalize sums.
            constexpr int PCP = WTN + 8;
            static_assert(WTN == 64, "wave tile width");
            f16* patch = (f16*)(smem + C_OFF) + wave * (32 * PCP);
            const bool relu_now = d.relu && !d.res;
            const bool has_res = d.res != nullptr && !(d.dbg & 16);
            const bool dense = d.osy == 1 && d.osx == 1 && d.OHg == d.OH && d.OWg == d.OW;
            int fr_e = fr, fh_e = fh, lane_e = lane;            // (opaque copies: keeps the epilogue's addresses out of the
            asm volatile("" : "+v"(fr_e), "+v"(fh_e), "+v"(lane_e));  //  persistent loop's invariant set, where they would spill)
            const int ch = lane_e & 7;
            const int col = cur.tile_n * BN + wn * WTN + ch * 8;
            int ct_ph = 0, ct_co = 0;
            if (CTF) gdt_ctf_column(col, d.phase_cout, ct_ph, ct_co);
            float st1[8], st2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }
            float4 bvs[TN][4];                                   // bias of this lane's channels, fetched once per tile
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bvs[j][g] = d.bias ? *(const float4*)(d.bias + cur.tile_n * BN + wn * WTN + j * 32 + 8 * g + 4 * fh_e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bv = bvs[j][g];
                        const f32x16& a = acc[i][j];
                        float v0 = a[4 * g] + bv.x, v1 = a[4 * g + 1] + bv.y, v2 = a[4 * g + 2] + bv.z, v3 = a[4 * g + 3] + bv.w;
                        if (relu_now) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        f16x4 h; h[0] = (f16)v0; h[1] = (f16)v1; h[2] = (f16)v2; h[3] = (f16)v3;
                        *(f16x4*)(patch + fr_e * PCP + j * 32 + 8 * g + 4 * fh_e) = h;
                    }
                unsigned offs[4];
                f16x8 rv[4];
                unsigned okmask = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = cur.tile_m * BM + wm * WTM + i * 32 + (lane_e >> 3) + 8 * q;
                    const bool ok = (m < d.M) & (col < d.Cout);
                    unsigned pix = (unsigned)m;                              // dense output grid: the GEMM row IS the pixel
                    if (CTF || !dense) {                                     // (transposed convs: every other row / column)
                        const int mm = m < d.M ? m : 0;
                        const int n = mm / hw_g, rem = mm - n * hw_g;
                        const int oy = rem / d.OWg, ox = rem - oy * d.OWg;
                        pix = CTF ? (unsigned)((n * d.OH + 2 * oy + (ct_ph >> 1)) * d.OW + 2 * ox + (ct_ph & 1))
                                  : (unsigned)((n * d.OH + oy * d.osy + d.ooy) * d.OW + ox * d.osx + d.oox);
                    }
                    offs[q] = ok ? (CTF ? pix * (unsigned)d.phase_cout + ct_co : pix * (unsigned)d.Cout + col) : 0u;
                    okmask |= (ok ? 1u : 0u) << q;
                    if (has_res) rv[q] = *(const f16x8*)(d.res + offs[q]);     // offset 0 is a valid address for masked pieces
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f16x8 v = *(const f16x8*)(patch + ((lane_e >> 3) + 8 * q) * PCP + ch * 8);
                    if (d.stats && ((okmask >> q) & 1u)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; st1[e] += f; st2[e] += f * f; }
                    }
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float t = (float)v[e] + (float)rv[q][e];
                            if (d.relu) t = fmaxf(t, 0.f);
                            v[e] = (f16)t;
                        }
                    }
                    if (((okmask >> q) & 1u) && !(d.dbg & 8)) *(f16x8*)(d.out + offs[q]) = v;
                }
            }
            if (d.stats) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int msk = 8; msk < 64; msk <<= 1) { st1[e] += __shfl_xor(st1[e], msk); st2[e] += __shfl_xor(st2[e], msk); }
                    if (CTF) { st1[e] += __shfl_xor(st1[e], 4); st2[e] += __shfl_xor(st2[e], 4); }     // the wave's two sub-pixel phases of a channel
                }
                constexpr int WPR = WGM / 2;                         // wave rows per 128-row record
                const int rec = cur.tile_m * 2 + wm / WPR;
                if (!CTF && lane_e < 8 && col < d.Cout && rec * 128 < d.M) {
                    float* dst = d.stats + ((long)(d.stats_tile_base + (wm % WPR) * (d.M / 128) + rec) * 2) * d.Cout + col;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { dst[e] = st1[e]; dst[d.Cout + e] = st2[e]; }
                }
                if (CTF && lane_e < 4 && rec * 128 < d.M) {          // one record set per phase pair (gdt_ctf_column)
                    const int pair = ((cur.tile_n * WGN + wn) / (d.phase_cout >> 5)) & 1;
                    float* dst = d.stats + ((long)(pair * (d.M / 128) + rec) * 2) * d.phase_cout + ct_co;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { dst[e] = st1[e]; dst[d.phase_cout + e] = st2[e]; }
                }
            }
        }
        if (!nxt.valid) break;
        cur = nxt; vb += gridDim.x;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) load_b(kk, cur.tile_n, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) afr[0][i] = a_frag(so, i, 0);
    }
}

template <int BN, int WGM, int WGN, bool NORM, bool CTF = false>
int launch_irb(const ConvLaunch& d, hipStream_t stream) {
    constexpr size_t lds = irb_lds_bytes<BN, WGM>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_igemm_rb_kernel<BN, WGM, WGN, NORM, CTF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    const int vblocks = gdt_grid_for_tiles((d.M + BM - 1) / BM, d.CoutPad / BN);
    const int grid = vblocks < cus ? vblocks : cus;
    hipLaunchKernelGGL((conv_igemm_rb_kernel<BN, WGM, WGN, NORM, CTF>), dim3(grid), dim3(WGM * WGN * 64), lds, stream, d, vblocks);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// Eligible: fp16 NHWC in and out, Cin a multiple of 64, fragment-ordered weights, at least two K-steps (the A tile is staged
// two steps ahead), tensors addressable with 32-bit element offsets, enough tiles to fill the chip; with a folded
// InstanceNorm additionally Cin <= 256 and whole tiles inside one image.
// record sets written for a launch with fused statistics: waves whose WTM rows are a fraction of a 128-row record write to
// separate sets (BN 256: 1, 128: 2, 64: 4); the fused transposed form writes one per phase pair
int gdt_conv_igemm_rb_stats_sets(const ConvLaunch& d) {
    if (d.phase_cout) return 2;
    return d.CoutPad % 256 == 0 ? 1 : (d.CoutPad % 128 == 0 ? 2 : 4);
}

bool gdt_conv_igemm_rb_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_IRB"); return e ? atoi(e) : 1; }();   // 0 off, 2 force
    if (mode == 0 || !d.w_frag || d.out_f32 || !d.out || d.Cin % 64 != 0 || d.Kpad != d.ntaps * d.Cin || d.Kpad < 128) return false;
    if (d.CoutPad % 64 != 0 || d.Cout % 8 != 0 || d.in_res || d.in_out) return false;
    if (d.phase_cout && (d.CoutPad % 256 != 0 || 256 % d.phase_cout != 0 || d.phase_cout < 64 || d.Cout != 4 * d.phase_cout || d.res || d.M % 128 != 0)) return false;
    if ((long)d.N * d.H * d.W * d.Cin >= (1L << 32) || (long)d.N * d.OH * d.OW * (d.phase_cout ? d.phase_cout : d.Cout) >= (1L << 32)) return false;
    if (d.stats && ((d.OHg * d.OWg) % 128 != 0 || d.CoutPad % 128 != 0 || d.M % 128 != 0)) return false;    // (64-wide tiles with statistics: conv_igemm.hip measured faster)
    if (d.in_norm && (d.Cin > 256 || (d.OHg * d.OWg) % BM != 0)) return false;
    if (mode == 2) return true;
    // History (sustained bench.py A/B runs on 1x MI355X): with the whole-tile LDS transpose and its barriers this kernel tied
    // with conv_igemm.hip on the plain 1x1 convs of ResNet-101 (1526 vs 1545 descriptors/s) and was used only where it also
    // folds an InstanceNorm; with the wave-private epilogue it is ahead there too (1777 vs 1671).
    const int bn = d.CoutPad % 256 == 0 ? 256 : (d.CoutPad % 128 == 0 ? 128 : 64);
    static const int min_tiles = [] { const char* e = getenv("GDT_IRB_MIN_TILES"); return e ? atoi(e) : 256; }();
    return (long)((d.M + BM - 1) / BM) * (d.CoutPad / bn) >= min_tiles;
}

int gdt_launch_conv_igemm_rb(const ConvLaunch& d_in, hipStream_t stream, int* variant) {
    static const int dbg = [] { const char* e = getenv("GDT_IRB_DBG"); return e ? atoi(e) : 0; }();
    ConvLaunch d = d_in;
    d.dbg = dbg;
    const int bn = d.CoutPad % 256 == 0 ? 256 : (d.CoutPad % 128 == 0 ? 128 : 64);
    *variant = 940000 + bn;
    if (d.phase_cout) return d.in_norm ? launch_irb<256, 2, 4, true, true>(d, stream) : launch_irb<256, 2, 4, false, true>(d, stream);
    if (d.in_norm) {
        if (bn == 256) return launch_irb<256, 2, 4, true>(d, stream);
        if (bn == 128) return launch_irb<128, 4, 2, true>(d, stream);
        return launch_irb<64, 8, 1, true>(d, stream);
    }
    if (bn == 256) return launch_irb<256, 2, 4, false>(d, stream);
    if (bn == 128) return launch_irb<128, 4, 2, false>(d, stream);
    return launch_irb<64, 8, 1, false>(d, stream);
}
