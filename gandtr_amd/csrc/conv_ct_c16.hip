// ConvTranspose2d(k3, s2, p1, op1) of the "f16c" precision mode (the generator's two up-sampling layers, p2p_networks.py:289-300) on the
// 16 x 16 MFMA shapes -- the skeleton of conv3x3_halo_c16.hip (persistent workgroups, 16 x 16 patch of INPUT pixels, fp32 halo read through
// registers with the producer's InstanceNorm (+ReLU / + residual) applied in fp32, fp16 + fp4 planes in LDS, weights streamed L2 ->
// registers as 15 KB records, inline-asm MFMAs tied to AGPRs, a hand-laid loop with the staging in phases), for the layer the 32 x 32 kernel
// (conv3x3_halo_c.hip, FORM 1) runs at 0.14 of the fp16 peak.
//
// GEMM view: rows = input pixels; columns = 4 sub-pixel phases x cout; K = 4 input shifts (dy, dx) in {0, 1}^2 x cin, where the (shift, phase)
// pairs a 3 x 3 kernel with stride 2 does not produce are zero blocks: phase (py, px) takes shift (dy, dx) iff (dy <= py) and (dx <= px) --
// 9 of 16.  out[2y + py][2x + px][co] = sum over those shifts of W . in[y + dy][x + dx].
//   * a wave's 64 columns = the FOUR PHASES of 16 output channels: channel block cb of the wave IS phase cb, so which blocks a shift feeds is a
//     compile-time constant (shift 0: all four, 1: phases 1 3, 2: phases 2 3, 3: phase 3): the zero blocks cost neither MFMAs nor weight loads,
//     and every wave has the same work.  The four waves take 64 output channels; Cout = 128 is two column tiles.
//   * 17 x 17 halo (shifts 0 / +1), zero beyond the image.  LDS rows, swizzles and the pixel <-> MFMA column map as in conv3x3_halo_c16.hip
//     (17-pixel rows keep its bank argument: a lane group's eight even and eight odd pixels fall into different bank halves).
//   * THREE halo rounds in flight per thread (24 KB per workgroup): these layers move 1.1-1.6 GB for 1.2 GMAC per image -- their floor is HBM
//     (0.2 / 0.3 ms at batch 64), and with one 8 KB round in flight a workgroup draws ~4 B/clk.  The staging phases (one per patch row of MFMAs)
//     sit in the slots of shifts 0-2; shift 3's slots (one MFMA per activation fragment) carry none.
//   * epilogue straight from the accumulators: block cb of pixel (y, x) goes to output pixel (2y + py, 2x + px), 16 bytes per lane, 64 contiguous
//     bytes per pixel and instruction.  Statistics over the wave's 256 pixels x 4 phases per output channel, one record per wave.
// Whole 16 x 16 patches, cout 64 or 128, no epilogue residual, no write-back; everything else stays on conv3x3_halo_c.hip.
#include <cstdio>
#include <cstdlib>

#include <vector>

#include "gdt_common.h"

#ifndef GDT_CT16_ABL
#define GDT_CT16_ABL 0          // timing-only ablations: 1 no halo staging   2 no MX MFMAs / loads
#endif

namespace {

constexpr int ROWB = 128, QROWB = 64;
constexpr int HW_ = 17, HROWS = HW_ * HW_, HROWS_PAD = 296;
constexpr int A_BYTES = HROWS_PAD * ROWB;                  // 37888
constexpr int Q_BYTES = HROWS_PAD * QROWB;                 // 18944
constexpr int STAGE_BYTES = A_BYTES + Q_BYTES;             // 56832
constexpr int NORM_BYTES = 4096 + 64;
constexpr size_t LDS_BYTES = 2 * (size_t)STAGE_BYTES + NORM_BYTES;
constexpr int NT = 256, RPR = NT / 8, NR = (HROWS_PAD + RPR - 1) / RPR;      // 10 rounds of 32 halo rows per chunk
constexpr int SDIST = 3;                                   // halo rounds in flight per thread
constexpr int NPHASE = 14;                                 // staging phases per round: 10 of the store, 4 of the load
constexpr int NQ = NR * NPHASE;                            // 140 phases per chunk, one per usable position (9 slots of 16 patch rows)
static_assert(NQ <= 9 * 16, "staging phases of a chunk fit the slots of shifts 0-2");

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TileAt { int n, y0, x0, tile_m, tile_n; bool valid; };

__device__ __forceinline__ void mfma16(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_mx(f32x4& acc, const v6i& a6, const v4i& b4, int sa, int sb) {
    asm("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:4" : "+a"(acc) : "v"(a6), "v"(b4), "v"(sa), "v"(sb));
}

// channel blocks (= phases) that input shift t = dy * 2 + dx feeds
__host__ __device__ constexpr unsigned cb_mask(int t) { return t == 0 ? 0xFu : (t == 1 ? 0xAu : (t == 2 ? 0xCu : 0x8u)); }
__host__ __device__ constexpr bool cb_on(int t, int cb) { return ((cb_mask(t) >> cb) & 1u) != 0; }

// MODE bits: 1 = the producer's InstanceNorm (+ReLU) is applied while staging; 2 = ... plus a residual
template <int MODE>
__global__ __launch_bounds__(NT) void conv_ct_c16_kernel(const ConvLaunch d, const int vblocks) {
    constexpr bool NORM = (MODE & 1) != 0, RES = (MODE & 2) != 0;
    constexpr int RING = 4;                  // fp16 weight ring, in half-steps (8 per chunk)
#ifndef GDT_CT16_AW
#define GDT_CT16_AW 8
#endif
    constexpr int AW = GDT_CT16_AW, QW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // nothing that only depends on the lane id stays live across the chunk loop (the allocator spills it, and a scratch reload is a vector-memory
    // wait behind three rounds of halo loads): such values are re-made from v_mbcnt where they are used
    auto lane_now = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    const float* __restrict__ inf = (const float*)d.in;
    const float* __restrict__ resf = (const float*)d.in_res;
    const int pc = d.phase_cout;             // real output channels

    const int tiles_x = d.W >> 4, tiles_y = d.H >> 4;
    const int tpi = tiles_x * tiles_y, ntm = d.N * tpi, ntn = pc >> 6;      // a column tile = 64 output channels x 4 phases
    auto tile_at = [&](int vb) -> TileAt {
        TileAt t;
        t.valid = vb < vblocks && gdt_tile_of_block(vb, ntm, ntn, t.tile_m, t.tile_n);
        if (!t.valid) { t.tile_m = 0; t.tile_n = 0; }
        t.n = t.tile_m / tpi;
        const int tr = t.tile_m - t.n * tpi;
        t.y0 = (tr / tiles_x) << 4; t.x0 = (tr % tiles_x) << 4;
        return t;
    };
    int vb = blockIdx.x;
    TileAt cur = tile_at(vb);
    if (!cur.valid) return;
    // Every workgroup runs the same number of equally long tiles; left alone all 256 reach their epilogues together, the chip alternates between a
    // read phase at a fraction of the HBM rate and a write burst (256 KB per workgroup: these layers write 2 bytes per byte read), and each wave
    // sits behind its own stores (vmcnt retires in order).  Four phase groups, d.stagger_us apart, let one group's writes meet the others' reads.
    if (d.stagger_us > 0) {
        const int grp = (blockIdx.x >> 3) & 3;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();            // 100 MHz
        const unsigned long long wait = (unsigned long long)grp * d.stagger_us * 100;
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }

    // ---- halo loader (conv3x3_halo_c16.hip), shifts 0 / +1: halo pixel (hy, hx) = input pixel (y0 + hy, x0 + hx), zero beyond the image
    const float lo_scale = __builtin_ldexpf(1.f, -d.c_lo_exp), hi_scale = __builtin_ldexpf(1.f, d.c_hi_exp);
    struct Pend { float4 r0, r1, s0, s1; unsigned goff; bool ok; };
    Pend pendv[SDIST];
    auto load_piece_part = [&](Pend& pend, const TileAt& ta, int chunk, int r, int part) {
        if (part == 0) {
            const int ln = lane_now();
            const int lr = wave * 8 + (ln >> 3);
            const int h = min(r * RPR + lr, HROWS_PAD - 1);
            const int hy = (h * 3856) >> 16, hx = h - hy * HW_;
            const int iy = ta.y0 + hy, ix = ta.x0 + hx;
            const int ry = min(iy, d.H - 1), rx = min(ix, d.W - 1);
            pend.goff = (((unsigned)((ta.n * d.H + ry) * d.W + rx) << (d.lc8 + 5)) + (chunk * 8 + (ln & 7)) * 32);
            pend.ok = (h < HROWS) & (iy < d.H) & (ix < d.W);
            pend.r0 = *(const float4*)((const char*)inf + pend.goff);
        }
        if (part == 1) pend.r1 = *(const float4*)((const char*)inf + pend.goff + 16);
        if (RES && part == 2) pend.s0 = *(const float4*)((const char*)resf + pend.goff);
        if (RES && part == 3) pend.s1 = *(const float4*)((const char*)resf + pend.goff + 16);
    };
    float* nlds = (float*)(smem + 2 * STAGE_BYTES);
    auto stage_norm = [&](const TileAt& ta, int slot) {
        for (int i = tid; i < d.Cin / 2; i += NT) {              // float4 = 2 channels x (mean, rstd) -> (scale, shift)
            const float4 v = *(const float4*)(d.in_norm + (long)ta.n * d.Cin * 2 + i * 4);
            *(float4*)(nlds + slot * 512 + i * 4) = make_float4(v.y, -v.x * v.y, v.w, -v.z * v.w);
        }
    };
    float4 nf[4];                            // (scale, shift) of the lane's 8 channels in the chunk being staged
    auto load_nf = [&](int slot, int chunk) {
        if (!NORM) return;
        const float4* np4 = (const float4*)(nlds + slot * 512 + (chunk * 8 + (lane_now() & 7)) * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) nf[k] = np4[k];
    };
    float sa[8];
    unsigned sou[4], sqlo = 0, sqhi = 0;
#define GDT_PIN4(v) asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w))
#define GDT_PIN8(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
    // store of round r in phases 0-9 (conv3x3_halo_c16.hip: an empty asm volatile on a phase's inputs pins its arithmetic to the phase)
    auto store_phase = [&](Pend& pend, int stage_off, int r, int ph) {
        if (ph == 0) {
            GDT_PIN4(pend.r0); GDT_PIN4(pend.r1);
            sa[0] = pend.r0.x; sa[1] = pend.r0.y; sa[2] = pend.r0.z; sa[3] = pend.r0.w; sa[4] = pend.r1.x; sa[5] = pend.r1.y; sa[6] = pend.r1.z; sa[7] = pend.r1.w;
        }
        if (ph >= 1 && ph <= 9) GDT_PIN8(sa);
        if (NORM && (ph == 0 || ph == 1)) {
            const float lo = d.in_relu ? 0.f : -3.0e38f;
#pragma unroll
            for (int k = 2 * ph; k < 2 * ph + 2; ++k) {
                const float4 v = nf[k];
                sa[2 * k] = fmaxf(fmaf(sa[2 * k], v.x, v.y), lo);
                sa[2 * k + 1] = fmaxf(fmaf(sa[2 * k + 1], v.z, v.w), lo);
            }
        }
        if (NORM && RES && ph == 2) {
            GDT_PIN4(pend.s0); GDT_PIN4(pend.s1);
            sa[0] += pend.s0.x; sa[1] += pend.s0.y; sa[2] += pend.s0.z; sa[3] += pend.s0.w;
            sa[4] += pend.s1.x; sa[5] += pend.s1.y; sa[6] += pend.s1.z; sa[7] += pend.s1.w;
        }
        if (ph == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sa[e] = pend.ok ? sa[e] : 0.f;
            sqlo = 0; sqhi = 0;
        }
#define GDT_Q4P(k)                                                                                                                   \
        if (ph == 5 + k) {                                                                                                           \
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(sou[k]) : "v"(sa[2 * k]), "v"(sa[2 * k + 1]));                                  \
            float l0, l1;                                                                                                            \
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(sou[k]), "v"(sa[2 * k]));         \
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(sou[k]), "v"(sa[2 * k + 1]));     \
            sqlo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(sqlo, l0, l1, lo_scale, k);                                              \
            sqhi = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(sqhi, __builtin_bit_cast(f16x2, sou[k]), hi_scale, k);                   \
        }
        GDT_Q4P(0) GDT_Q4P(1) GDT_Q4P(2) GDT_Q4P(3)
#undef GDT_Q4P
        if (ph == 9) {
            const int ln = lane_now();
            const int row = min(r * RPR + wave * 8 + (ln >> 3), HROWS_PAD - 1);
            const int phy = (row * 3856) >> 16, phx = row - phy * HW_;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 ov = {sou[0], sou[1], sou[2], sou[3]};
            const int q = ln & 7;
            *(f16x8*)(smem + stage_off + row * ROWB + ((q ^ ((phx >> 1) & 7)) << 4)) = __builtin_bit_cast(f16x8, ov);
            const int key2 = (phx >> 2) & 3;
            const int qo = stage_off + A_BYTES + row * QROWB + ((((q >> 2) << 1) ^ key2) << 4) + ((q & 3) << 2);
            *(unsigned*)(smem + qo) = sqlo;
            *(unsigned*)(smem + (qo ^ 16)) = sqhi;
        }
    };
#undef GDT_PIN4
#undef GDT_PIN8
    // staging phase q (0 .. NQ-1) of the chunk being staged: rounds are loaded SDIST rounds ahead of their store --
    // L0 L1 L2 [S0 L3] [S1 L4] ... [S6 L9] S7 S8 S9, a load = 4 phases, a store = 10
    auto stage_q = [&](const TileAt& ta, int chunk, int stage_off, int q) {
        if ((GDT_CT16_ABL & 1) || q >= NQ) return;
        if (q < 4 * SDIST) { load_piece_part(pendv[(q / 4) % SDIST], ta, chunk, q / 4, q % 4); return; }
        const int qq = q - 4 * SDIST;
        if (qq < (NR - SDIST) * NPHASE) {
            const int b = qq / NPHASE, i = qq % NPHASE;
            if (i < 10) store_phase(pendv[b % SDIST], stage_off, b, i);
            else load_piece_part(pendv[(b + SDIST) % SDIST], ta, chunk, b + SDIST, i - 10);
            return;
        }
        const int q3 = qq - (NR - SDIST) * NPHASE;
        store_phase(pendv[(NR - SDIST + q3 / 10) % SDIST], stage_off, NR - SDIST + q3 / 10, q3 % 10);
    };

    // ---- weights: one 15 KB record per (64 columns, 64 k-values) (conv3x3_halo_c16.hip), k = shift * cin + c; only the blocks a shift feeds are fetched
    constexpr long WREC = 15360;
    auto wgrp = [&](int tile_n) -> long { return (long)tile_n * 4 + wave; };
    const int nms = d.Kpad >> 6, cin64 = d.Cin >> 6;
    f16x8 bw[RING][4];
    v6i bq[4];
    v4i bqs;
    auto lane_bytes = [&](int per_lane) -> unsigned {
        unsigned v = lane_now() * per_lane;
        asm volatile("" : "+v"(v));
        return v;
    };
    unsigned lo16 = lane_bytes(16), lo8 = lane_bytes(8);
    auto load_bw = [&](int rs, int cb, int tile_n, long ks) {        // ks: uniform index of the 32-k step
        const char* wb = (const char*)d.w_c16 + (wgrp(tile_n) * nms + (ks >> 1)) * WREC + (ks & 1) * 4096;
        bw[rs][cb] = *(const f16x8*)(wb + lo16 + cb * 1024);
    };
    auto load_bq_part = [&](int part, int tile_n, long ms) {
        const char* rec = (const char*)d.w_c16 + (wgrp(tile_n) * nms + ms) * WREC;
        const int cb = part >> 1;
        if (part == 8) bqs = *(const v4i*)(rec + 14336 + lo16);
        else if ((part & 1) == 0) {
            const v4i qa = *(const v4i*)(rec + 8192 + lo16 + cb * 1024);
            bq[cb] = __builtin_shufflevector(__builtin_shufflevector(qa, qa, 0, 1, 2, 3, -1, -1), bq[cb], 0, 1, 2, 3, 10, 11);
        } else {
            const v2i qb = *(const v2i*)(rec + 12288 + lo8 + cb * 512);
            bq[cb] = __builtin_shufflevector(bq[cb], __builtin_shufflevector(qb, qb, 0, 1, -1, -1, -1, -1), 0, 1, 2, 3, 6, 7);
        }
    };

    // ---- activation fragment addresses (conv3x3_halo_c16.hip): lane (n, g) = pixel x = PIX(n) of a patch row, k-slot g
    const int fn = lane & 15, fg = lane >> 4;
    const int px = fn < 4 ? 2 * fn : (fn < 12 ? 2 * (fn - 4) + 1 : 2 * (fn - 8));
    int vt[2], vq[2];
#pragma unroll
    for (int tx = 0; tx < 2; ++tx) {
        vt[tx] = px * ROWB + ((fg ^ (((px + tx) >> 1) & 7)) << 4);
        vq[tx] = A_BYTES + px * QROWB + ((fg ^ (((px + tx) >> 2) & 3)) << 4);
    }
    auto a_frag = [&](int pb, int t, int s) -> f16x8 {       // shift t = dy * 2 + dx
        return *(const f16x8*)(smem + (vt[t & 1] ^ (s << 6)) + ((pb + (t >> 1)) * HW_ + (t & 1)) * ROWB);
    };
    auto a_qfrag = [&](int pb, int t) -> v4i {
        return *(const v4i*)(smem + vq[t & 1] + ((pb + (t >> 1)) * HW_ + (t & 1)) * QROWB);
    };
    auto flip_stage = [&](int delta) {
#pragma unroll
        for (int k = 0; k < 2; ++k) { vt[k] += delta; vq[k] += delta; }
    };
    const int a_scale = (fg & 1) ? 127 + d.c_hi_exp : 127 - d.c_lo_exp;

    const int nchunks = d.Cin >> 6;
    // half-step u = 2 t + s of a chunk; u >= 8: the first half-steps of the chunk staged now
    // ---- prologue
#pragma unroll
    for (int u = 0; u < RING - 1; ++u)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            if (cb_on(u >> 1, cb)) load_bw(u, cb, cur.tile_n, (long)(((u >> 1) * cin64) * 2 + (u & 1)));
    if (NORM) {
        stage_norm(cur, 0);
        __syncthreads();
    }
    load_nf(0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int part = 0; part < 4; ++part) load_piece_part(pendv[0], cur, 0, r, part);
#pragma unroll
        for (int ph = 0; ph < 10; ++ph) store_phase(pendv[0], 0, r, ph);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SDIST; ++k) {
#pragma unroll
        for (int part = 0; part < 4; ++part) load_piece_part(pendv[k], cur, 0, 0, part);      // (placeholder values: overwritten before their first use)
    }

    f16x8 afr[AW];
    v4i aq[QW];
#pragma unroll
    for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0);

    int so = 0, slot = 0;
#ifdef GDT_C_STAMP
    unsigned long long st_body = 0, st_cbar = 0, st_tbar = 0, st_epi = 0, st_t = __builtin_amdgcn_s_memtime(), st_n = 0;
    const unsigned long long st_begin = st_t;
#define GDT_STAMP(acc_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - st_t; st_t = now_; }
#else
#define GDT_STAMP(acc_)
#endif
    for (;;) {
        const TileAt nxt = tile_at(vb + gridDim.x);
        f32x4 acc[16][4];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

        for (int c = 0; c < nchunks; ++c) {
            const bool last = c + 1 == nchunks;
            const bool to_next = last && nxt.valid;
            const TileAt sta = to_next ? nxt : cur;
            const int sc = last ? 0 : c + 1, sslot = to_next ? slot ^ 1 : slot;
            if (NORM && nxt.valid && c == nchunks - 2) stage_norm(nxt, slot ^ 1);
            load_nf(sslot, sc);
            lo16 = lane_bytes(16); lo8 = lane_bytes(8);
            auto ks_of = [&](int u) -> long { return u < 8 ? (long)(((u >> 1) * cin64 + c) * 2 + (u & 1)) : (long)((((u - 8) >> 1) * cin64 + sc) * 2 + (u & 1)); };
            auto tn_of = [&](int u) -> int { return (u >= 8 && last) ? nxt.tile_n : cur.tile_n; };
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // staging position of patch row pb of this shift's slot sl (0 .. 2): shifts 0-2 only
                auto qpos = [&](int sl, int pb) -> int { return t < 3 ? (t * 3 + sl) * 16 + pb : NQ; };
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int u = 2 * t + s;
#pragma unroll
                    for (int pb = 0; pb < 16; ++pb) {
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            if (cb_on(t, cb)) mfma16(acc[pb][cb], bw[u % RING][cb], afr[pb % AW]);
                        if (pb + AW < 16) afr[pb % AW] = a_frag(pb + AW, t, s);
                        else if (s == 0) afr[pb % AW] = a_frag(pb + AW - 16, t, 1);
                        else if (t < 3) afr[pb % AW] = a_frag(pb + AW - 16, t + 1, 0);
                        // weights of half-step u + RING - 1 (ring slot of half-step u - 1), one block per fourth patch row
                        if ((pb & 3) == 2) {
                            const int un = u + RING - 1, cbn = pb >> 2;
                            if (cb_on((un & 7) >> 1, cbn)) load_bw(un % RING, cbn, tn_of(un), ks_of(un));
                        }
                        // MX weights of this shift (behind the previous shift's MX run): one part per second patch row of the first half-step
                        if (!(GDT_CT16_ABL & 2) && (pb & 1) == 1 && (s == 0 || pb == 1)) {
                            const int part = s == 0 ? pb >> 1 : 8;
                            if (part == 8 || cb_on(t, part >> 1)) load_bq_part(part, cur.tile_n, (long)(t * cin64 + c));
                        }
                        if (!(GDT_CT16_ABL & 2) && s == 1 && pb >= 16 - QW) aq[pb - (16 - QW)] = a_qfrag(pb - (16 - QW), t);
                        stage_q(sta, sc, STAGE_BYTES - so, qpos(s, pb));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int pb = 0; pb < 16; ++pb) {
                    if (!(GDT_CT16_ABL & 2)) {
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            if (cb_on(t, cb)) mfma16_mx(acc[pb][cb], bq[cb], aq[pb % QW], bqs[cb], a_scale);
                        if (pb + QW < 16) aq[pb % QW] = a_qfrag(pb + QW, t);
                    }
                    stage_q(sta, sc, STAGE_BYTES - so, qpos(2, pb));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            GDT_STAMP(st_body)
            if (!last) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                GDT_STAMP(st_cbar)
                flip_stage(STAGE_BYTES - 2 * so);
                so = STAGE_BYTES - so;
#pragma unroll
                for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0);
            }
        }

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        GDT_STAMP(st_tbar)
        const int so_next = STAGE_BYTES - so;

        // ------------------------------------------------------------ epilogue: block cb = phase (py, px) of 16 output channels
        {
            float* __restrict__ outp = (float*)d.out;
            const int lane_e = lane_now();
            const int n_e = lane_e & 15, g_e = lane_e >> 4;
            const int x_e = n_e < 4 ? 2 * n_e : (n_e < 12 ? 2 * (n_e - 4) + 1 : 2 * (n_e - 8));
            const int co = cur.tile_n * 64 + wave * 16 + 4 * g_e;             // this lane's 4 output channels (of every phase)
            // bias: per GEMM column of the 32 x 32 kernel's order (gdt_ctc_column); phase 0's column of channel co
            const float4 bv = d.bias ? *(const float4*)(d.bias + ((co >> 5) << 7) + (co & 31)) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float lo = d.relu ? 0.f : -__builtin_inff();
            unsigned o = (unsigned)((cur.n * d.OH + 2 * cur.y0) * d.OW + 2 * (cur.x0 + x_e)) * (unsigned)pc + (unsigned)co;
            const unsigned rowstep = 2u * (unsigned)d.OW * (unsigned)pc;      // one input row down = two output rows
            const unsigned ophase[4] = {0u, (unsigned)pc, (unsigned)d.OW * (unsigned)pc, (unsigned)(d.OW + 1) * (unsigned)pc};
            float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int pb = 0; pb < 16; ++pb) {
                asm volatile("" : "+v"(o));
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const f32x4& a = acc[pb][cb];
                    float4 t = make_float4(a[0] + bv.x, a[1] + bv.y, a[2] + bv.z, a[3] + bv.w);
                    s1.x += t.x; s1.y += t.y; s1.z += t.z; s1.w += t.w;
                    s2.x += t.x * t.x; s2.y += t.y * t.y; s2.z += t.z * t.z; s2.w += t.w * t.w;
                    t.x = fmaxf(t.x, lo); t.y = fmaxf(t.y, lo); t.z = fmaxf(t.z, lo); t.w = fmaxf(t.w, lo);
                    *(float4*)(outp + o + ophase[cb]) = t;
                }
                o += rowstep;
            }
            if (d.stats) {
                auto merge = [](float v) -> float {
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));    // row_ror:8
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));    // row_ror:4
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));    // row_ror:2
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));    // row_ror:1
                    return v;
                };
                s1.x = merge(s1.x); s1.y = merge(s1.y); s1.z = merge(s1.z); s1.w = merge(s1.w);
                s2.x = merge(s2.x); s2.y = merge(s2.y); s2.z = merge(s2.z); s2.w = merge(s2.w);
                if (n_e == 0) {
                    // slab layout of the 32 x 32 kernel's transposed form: two records per patch, (sum, sum of squares) rows of phase_cout
                    float* dst = d.stats + ((long)(cur.tile_m * 2) * 2) * pc + co;
                    *(float4*)dst = s1;
                    *(float4*)(dst + pc) = s2;
                    *(float4*)(dst + 2l * pc) = make_float4(0.f, 0.f, 0.f, 0.f);
                    *(float4*)(dst + 3l * pc) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
#ifdef GDT_C_STAMP
#ifndef GDT_STAMP_NODRAIN
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (diagnostic only: the epilogue's stores are charged to the epilogue)
#endif
        GDT_STAMP(st_epi)
        ++st_n;
#endif
        if (!nxt.valid) break;
        cur = nxt; vb += gridDim.x; slot ^= 1;
        flip_stage(so_next - so);
        so = so_next;
#pragma unroll
        for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0);
    }
#ifdef GDT_C_STAMP
    if (lane == 0 && d.stamp_out) {
        unsigned long long* o = d.stamp_out + ((long)blockIdx.x * 4 + wave) * 8;
        o[0] = st_body; o[1] = st_cbar; o[2] = st_tbar; o[3] = st_epi; o[4] = __builtin_amdgcn_s_memtime() - st_begin; o[5] = st_n;
    }
#endif
}

template <int MODE>
int launch_ct16(const ConvLaunch& d, hipStream_t stream) {
    const int tiles = d.N * (d.W >> 4) * (d.H >> 4), ntn = d.phase_cout >> 6;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        GDT_CHECK_HIP(hipGetDevice(&dev));
        GDT_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        cus = cus / 8 * 8;
        GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_ct_c16_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    }
    const int vblocks = gdt_grid_for_tiles(tiles, ntn);
    const int grid = vblocks < cus ? vblocks : cus;
#ifdef GDT_C_STAMP
    static unsigned long long* stamp_buf = nullptr;
    static int stamp_calls = 0;
    ConvLaunch ds = d;
    if (!stamp_buf) GDT_CHECK_HIP(hipMalloc((void**)&stamp_buf, (size_t)cus * 4 * 8 * sizeof(unsigned long long)));
    GDT_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, (size_t)cus * 4 * 8 * sizeof(unsigned long long), stream));
    ds.stamp_out = stamp_buf;
    hipLaunchKernelGGL((conv_ct_c16_kernel<MODE>), dim3(grid), dim3(NT), LDS_BYTES, stream, ds, vblocks);
    if (++stamp_calls % 100 < 10) {
        GDT_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)grid * 4 * 8);
        GDT_CHECK_HIP(hipMemcpy(h.data(), stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s[6] = {0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)grid * 4; ++w) for (int k = 0; k < 6; ++k) s[k] += (double)h[w * 8 + k];
        const double nw = grid * 4.0, nt = s[5] / nw;
        fprintf(stderr, "[c stamp] MODE %d FORM 1 BN %d waves 4: tiles/wave %.1f; per tile: chunk bodies %.0f, chunk barriers %.0f, tile barrier %.0f, epilogue %.0f cycles; total per wave %.0f\n",
                MODE, d.Cin, nt, s[0] / nw / nt, s[1] / nw / nt, s[2] / nw / nt, s[3] / nw / nt, s[4] / nw);
    }
#else
    hipLaunchKernelGGL((conv_ct_c16_kernel<MODE>), dim3(grid), dim3(NT), LDS_BYTES, stream, d, vblocks);
#endif
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// Eligible: what the 32 x 32 kernel's transposed form takes (checked by the caller), whole 16 x 16 patches of input pixels, cout 64 or 128,
// 128 <= cin (two chunks at least: the folded norm's table of the next tile is staged one chunk ahead), the 16 x 16 records present.
bool gdt_conv_ct_c16_eligible(const ConvLaunch& d) {
    // OFF by default: correct (tests/test_hip_f16c.py::test_ct_c16_*) but 11 % slower than the 32 x 32 form it was written to replace (0.44 + 0.52 vs
    // 0.39 + 0.47 ms at batch 64; stamped: per tile 22 k cycles of epilogue -- 256 KB of stores at the workgroup's HBM share -- behind a body whose
    // staged loads the MFMA-paced phases cannot keep three rounds deep).  GDT_CONV_CT_C16=1 selects it (read per call: tests toggle it).
    const char* e = getenv("GDT_CONV_CT_C16");
    const int mode = e ? atoi(e) : 0;
    if (mode == 0 || !d.w_c16 || !gdt_conv_halo_c_ct_eligible(d)) return false;
    if ((d.H & 15) || (d.W & 15) || (d.phase_cout != 64 && d.phase_cout != 128) || d.Cin < 128 || d.in_out || d.res) return false;
    const long tiles = (long)d.N * (d.W >> 4) * (d.H >> 4) * (d.phase_cout >> 6);
    return tiles >= 128;
}

int gdt_launch_conv_ct_c16(const ConvLaunch& d_in, hipStream_t stream) {
    static const int stagger = [] { const char* e = getenv("GDT_CT16_STAGGER_US"); return e ? atoi(e) : 0; }();
    ConvLaunch d = d_in;
    d.stagger_us = stagger;
    if (!d.in_norm) return launch_ct16<0>(d, stream);
    return d.in_res ? launch_ct16<3>(d, stream) : launch_ct16<1>(d, stream);
}
