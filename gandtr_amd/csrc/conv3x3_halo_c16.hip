// 3x3 / stride-1 / pad-1 convolution of the "f16c" precision mode (conv3x3_halo_c.hip explains the arithmetic: fp32 NHWC activations,
// a*w ~= fp16(a)*fp16(w) on the fp16 MFMA + one block-scaled fp4 x fp6 MFMA that carries both rounding residuals) on the 16 x 16 MFMA SHAPES:
//     v_mfma_f32_16x16x32_f16 (two per 64 k-values)  +  v_mfma_scale_f32_16x16x128_f8f6f4 (one per 64 k-values, four 32-wide K blocks:
//     [a_lo | a_hi | a_lo' | a_hi']_fp4 . [w_hi | w_lo | w_hi' | w_lo']_fp6 for the two halves of the 64 k-values).
//
// Why a second kernel for the same layer (ResnetBlock convs of the generator, p2p_networks.py:480-494): the chip runs this kernel against its
// power governor, not against its issue slots -- whatever the loop does, the socket sits at ~1.15 kW and the clock settles where that
// power is reached (conv3x3_halo_c: 63 % matrix-pipe busy at 1.94 GHz; a tighter loop returns as a lower clock).  Per FLOP the 16 x 16 x 32
// shape reads and writes HALF the accumulator registers of 32 x 32 x 16 (K = 32 per accumulation instead of 16) and the governor lets it
// clock higher: profiles/experiments/mfma_shape_probe.hip, this instruction mix, operands in registers: 1194 vs 1045 TFLOP/s
// (1.79 vs 1.56 GHz), fp16 only 1860 vs 1600.  Cycles per FLOP are the same for both shapes.
//
// Structure (shared with conv3x3_halo_c.hip: persistent workgroups walking an XCD-chunked tile list, 16 x 16 output patch x 256 output
// channels, 18 x 18 fp32 halo per 64-channel chunk read through registers, InstanceNorm (+ReLU / + residual / + write-back) of the producer
// applied in fp32 while staging (MODE bits), split into an fp16 plane (128-byte rows) and an fp4 plane (64-byte rows) in LDS, two stages;
// weights streamed L2 -> registers in fragment order; swapped MFMA operands D[cout][pixel]):
//   * 4 waves, one per SIMD, 1 x 4: a wave owns all 256 pixels x 64 output channels = 16 pixel blocks (the patch's rows) x 4 channel blocks
//     of 16 x 16, 256 accumulator registers;
//   * pixel <-> MFMA column: lane column n holds pixel x = PIX(n) of its patch row (even x on columns 0-3 and 12-15, odd x on 4-11), so that
//     the 16 lanes a ds_read_b128 serves per cycle read eight even and eight odd pixels: with the row swizzles of the planes every fragment read
//     is bank-conflict-free for all three tap columns (with the identity map two of the three collide two-way);
//   * an fp16 activation fragment (one patch row x 32 channels) feeds the wave's 4 channel blocks, a weight fragment (16 channels x 32 k) the
//     16 pixel blocks: the same operand bytes per FLOP as the 32 x 32 kernel's 1 x 4 layout;
//   * per tap and chunk: 2 x 64 fp16 MFMAs, then 64 MX MFMAs whose activation operand is ONE 16-byte read (a pixel's whole fp4 row);
//   * epilogue straight from the accumulators: a lane holds 4 consecutive output channels of one pixel per block = one 16-byte store; the
//     four lane groups of a pixel make 64 contiguous bytes per instruction, two blocks a 128-byte line.  No LDS transpose, no epilogue
//     patches.  InstanceNorm statistics: per lane over the 16 patch rows, then a DPP rotate butterfly over the 16 pixel lanes: fixed
//     order, deterministic; one 256-row record per wave (the second 128-row record of the slab layout is written as zeros).
// Full 16 x 16 patches and 256-column tiles only (the generator's resblocks at any batch that fills the chip); everything else stays on
// conv3x3_halo_c.hip.
#include <cstdio>
#include <cstdlib>

#include <vector>

#include "gdt_common.h"

#ifndef GDT_C16_SCHED
#define GDT_C16_SCHED 1         // 1: sched_group_barrier interleave (per MFMA: at most one LDS read, two VALU; a memory operation every fourth)
#endif
#ifndef GDT_C16_RING
#define GDT_C16_RING 3
#endif
#ifndef GDT_C16_FULL_LINES
#define GDT_C16_FULL_LINES 1    // epilogue: pixel halves exchange a channel block (DPP row_ror:8) so that every store instruction writes whole 128-byte lines
#endif
#ifndef GDT_C16_BQ_SETS
#define GDT_C16_BQ_SETS 1       // register sets of the MX weights: 2 = the set of tap t + 1 is fetched during tap t (measured: no faster -- the weight stream is
#endif                          // throughput-, not latency-bound -- and 28 registers that the second halo round in flight needs more)
#ifndef GDT_C16_ABL
#define GDT_C16_ABL 0           // timing-only ablations: 1 no halo staging   2 no MX MFMAs / loads   4 no fp16 weight re-loads   8 no fp4 fragment re-loads   16 no MX weight re-loads   64 MX weights fetched into an unused set   128 every weight fetch from one hot 7 KB window   256 no output stores
#endif

namespace {

constexpr int ROWB = 128;          // bytes per row of the fp16 plane (64 halves of K)
constexpr int QROWB = 64;          // bytes per row of the fp4 plane
constexpr int HW_ = 18, HROWS = HW_ * HW_, HROWS_PAD = 328;
constexpr int A_BYTES = HROWS_PAD * ROWB;                  // 41984
constexpr int Q_BYTES = HROWS_PAD * QROWB;                 // 20992
constexpr int STAGE_BYTES = A_BYTES + Q_BYTES;             // 62976
constexpr int NORM_BYTES = 4096 + 64;
constexpr size_t LDS_BYTES = 2 * (size_t)STAGE_BYTES + NORM_BYTES;
constexpr int NT = 256, RPR = NT / 8, NR = (HROWS_PAD + RPR - 1) / RPR;      // threads, halo rows per loader round, rounds per chunk (11)
constexpr int NTAP = 9, SLOTS = NTAP * 3;      // per tap: two fp16 half-steps + the MX run
constexpr int SPR = 2;                         // loader round r: loaded at slot SPR * r, written to LDS at slot SPR * (r + SDIST)
#ifndef GDT_C16_SDIST
#define GDT_C16_SDIST 2
#endif
constexpr int SDIST = GDT_C16_SDIST;           // rounds in flight per thread: a round is consumed SDIST * SPR slots (~2400 cycles at 2) after its loads --
                                               // with one, ~1300 cycles, every round began with a wait for HBM (13-25 k cycles per tile, stamped)
static_assert((NR + SDIST) * SPR <= SLOTS, "halo rounds are spread over the slots of the previous chunk");

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TileAt { int n, y0, x0, tile_m, tile_n; bool valid; };

// The MFMAs as inline asm with the accumulator tied to an AGPR tuple ("+a"): for the four-pass 16 x 16 shapes the compiler does not tie vdst
// to src2, lets the 64 accumulator tuples wander through the 256 AGPRs -- all of which they occupy -- and moves them through VGPRs and scratch
// around every MFMA (first build: 1580 v_accvgpr_read + 1520 v_accvgpr_write + 250 scratch operations per chunk).  Operands arrive from LDS /
// global loads (the compiler's s_waitcnt covers asm operands); the accumulators are read by VALU only behind the tile-end barrier, far
// beyond the MFMA -> VALU wait states nothing pads inside asm.
__device__ __forceinline__ void mfma16(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_mx(f32x4& acc, const v6i& a6, const v4i& b4, int sa, int sb) {      // A: 32 e2m3 values per lane, B: 32 e2m1
    asm("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:4" : "+a"(acc) : "v"(a6), "v"(b4), "v"(sa), "v"(sb));
}

// MODE bits: 1 = the producer's InstanceNorm (+ReLU) is applied while staging; 2 = ... plus a residual; 4 = the transformed tensor is written back
template <int MODE>
__global__ __launch_bounds__(NT) void conv3x3_halo_c16_kernel(const ConvLaunch d, const int vblocks) {
    constexpr bool NORM = (MODE & 1) != 0, RES = (MODE & 2) != 0, WB = (MODE & 4) != 0;
    constexpr int RING = GDT_C16_RING;
    static_assert(18 % RING == 0, "ring slot of a half-step must not depend on the chunk");
    constexpr int AW = 8, QW = 4;            // activation fragment windows (fp16 plane / fp4 plane): register sets re-loaded AW / QW pixel blocks ahead
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ inf = (const float*)d.in;
    const float* __restrict__ resf = (const float*)d.in_res;
    float* __restrict__ wbf = (float*)d.in_out;

    const int tiles_x = d.W >> 4, tiles_y = d.H >> 4;
    const int tpi = tiles_x * tiles_y, ntm = d.N * tpi, ntn = d.CoutPad >> 8;
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

    // ---- halo loader: through registers, branch-free (conv3x3_halo_c.hip)
    const int lrow = tid >> 3;
    const bool refl = d.pad_reflect != 0;
    const float lo_scale = __builtin_ldexpf(1.f, -d.c_lo_exp), hi_scale = __builtin_ldexpf(1.f, d.c_hi_exp);   // the converts divide by their scale
    struct Pend { float4 r0, r1, s0, s1; unsigned goff; bool ok; };
    auto load_piece = [&](const TileAt& ta, int chunk, int r) -> Pend {
        int lr = lrow;
        asm volatile("" : "+v"(lr));
        const int h = min(r * RPR + lr, HROWS_PAD - 1);
        const int hy = (h * 3641) >> 16, hx = h - hy * HW_;
        const int q = lane & 7;
        const int iy = ta.y0 - 1 + hy, ix = ta.x0 - 1 + hx;
        const int cbyte = (chunk * 8 + q) * 32;
        int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
        int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
        ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);
        const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
        Pend p;
        p.goff = (((unsigned)((ta.n * d.H + ry) * d.W + rx) << (d.lc8 + 5)) + cbyte);      // byte offset (< 2^32, checked on the host)
        p.ok = (h < HROWS) & (inb | refl);
        p.r0 = *(const float4*)((const char*)inf + p.goff); p.r1 = *(const float4*)((const char*)inf + p.goff + 16);
        if (RES) { p.s0 = *(const float4*)((const char*)resf + p.goff); p.s1 = *(const float4*)((const char*)resf + p.goff + 16); }
        return p;
    };
    // ... one load instruction per call (main loop): part 0 computes the address and fetches the first 16 bytes
    Pend pendv[SDIST];
    auto load_piece_part = [&](const TileAt& ta, int chunk, int r, int part) {
        Pend& pend = pendv[r % SDIST];
        if (part == 0) {
            int lr = lrow;
            asm volatile("" : "+v"(lr));
            const int h = min(r * RPR + lr, HROWS_PAD - 1);
            const int hy = (h * 3641) >> 16, hx = h - hy * HW_;
            const int iy = ta.y0 - 1 + hy, ix = ta.x0 - 1 + hx;
            int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
            int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
            ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);
            const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
            pend.goff = (((unsigned)((ta.n * d.H + ry) * d.W + rx) << (d.lc8 + 5)) + (chunk * 8 + (lane & 7)) * 32);
            pend.ok = (h < HROWS) & (inb | refl);
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
    float4 nf[4];
    auto load_nf = [&](int slot, int chunk) {
        if (!NORM) return;
        const float4* np4 = (const float4*)(nlds + slot * 512 + (chunk * 8 + (lane & 7)) * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) nf[k] = np4[k];
    };
    auto store_piece = [&](int stage_off, int r, const Pend& p) {
        const int row = min(r * RPR + lrow, HROWS_PAD - 1);
        const int phy = (row * 3641) >> 16, phx = row - phy * HW_;
        float a[8] = {p.r0.x, p.r0.y, p.r0.z, p.r0.w, p.r1.x, p.r1.y, p.r1.z, p.r1.w};
        if (NORM) {
            const float lo = d.in_relu ? 0.f : -3.0e38f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = nf[k];
                a[2 * k] = fmaxf(fmaf(a[2 * k], v.x, v.y), lo);
                a[2 * k + 1] = fmaxf(fmaf(a[2 * k + 1], v.z, v.w), lo);
            }
            if (RES) {
                a[0] += p.s0.x; a[1] += p.s0.y; a[2] += p.s0.z; a[3] += p.s0.w;
                a[4] += p.s1.x; a[5] += p.s1.y; a[6] += p.s1.z; a[7] += p.s1.w;
            }
        }
        if (WB) {      // every piece stores the value of its clamped source pixel: identical bits from neighbouring patches, no branch
            *(float4*)((char*)wbf + p.goff) = make_float4(a[0], a[1], a[2], a[3]);
            *(float4*)((char*)wbf + p.goff + 16) = make_float4(a[4], a[5], a[6], a[7]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = p.ok ? a[e] : 0.f;
        unsigned ou[4], qlo = 0, qhi = 0;
#define GDT_Q4(k)                                                                                                                    \
        {                                                                                                                            \
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ou[k]) : "v"(a[2 * k]), "v"(a[2 * k + 1]));                                     \
            float l0, l1;                                                                                                            \
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(ou[k]), "v"(a[2 * k]));            \
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(ou[k]), "v"(a[2 * k + 1]));        \
            qlo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(qlo, l0, l1, lo_scale, k);                                                \
            qhi = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(qhi, __builtin_bit_cast(f16x2, ou[k]), hi_scale, k);                       \
        }
        GDT_Q4(0) GDT_Q4(1) GDT_Q4(2) GDT_Q4(3)
#undef GDT_Q4
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 ov = {ou[0], ou[1], ou[2], ou[3]};
        const int q = lane & 7;
        // fp16 plane: source chunk q (8 channels) at 16-byte position q ^ key, key = (x >> 1) & 7
        *(f16x8*)(smem + stage_off + row * ROWB + ((q ^ ((phx >> 1) & 7)) << 4)) = __builtin_bit_cast(f16x8, ov);
        // fp4 plane: a row = [lo 0-31][hi 0-31][lo 32-63][hi 32-63], 16-byte position p at p ^ key2, key2 = (x >> 2) & 3; this thread holds
        // channels 8q .. 8q+7: dword q & 3 of the lo / hi part of half q >> 2
        const int key2 = (phx >> 2) & 3;
        const int qo = stage_off + A_BYTES + row * QROWB + ((((q >> 2) << 1) ^ key2) << 4) + ((q & 3) << 2);
        *(unsigned*)(smem + qo) = qlo;
        *(unsigned*)(smem + (qo ^ 16)) = qhi;
    };

    // The same work in PHASES for the main loop: the MFMAs there are inline asm, which the compiler neither schedules around nor sees as long
    // operations -- left alone it sinks every LDS read to just in front of its MFMA and runs the ~100 staging instructions of a round in one
    // piece.  So the loop body is laid out by hand: after the four MFMAs of a patch row comes one phase (<= 8 instructions) of the staging,
    // fenced by sched_barrier; a round's store (phases 0-9) and the next round's load (10, 11) share a slot of 16 patch rows.
    float sa[8];
    unsigned sou[4], sqlo = 0, sqhi = 0;
    auto stage_phase = [&](const TileAt& ta, int chunk, int stage_off, int sl, int ph) {
        if ((GDT_C16_ABL & 1) || sl % SPR != 0) return;
        const int r = sl / SPR;
        const bool st = r >= SDIST && r - SDIST < NR, ld = r < NR;
        Pend& pend = pendv[r % SDIST];      // (the round stored now and the round loaded behind it share a buffer)
        // (an empty asm volatile on a phase's inputs pins its arithmetic to the phase: pure VALU instructions carry no ordering against the
        //  sched_barrier fences when the block is linearised, and the compiler otherwise runs the normalisation right behind the loads -- with
        //  the wait for them)
#define GDT_PIN4(v) asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w))
#define GDT_PIN8(a) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
        if (st && ph == 0) {
            GDT_PIN4(pend.r0); GDT_PIN4(pend.r1);
            sa[0] = pend.r0.x; sa[1] = pend.r0.y; sa[2] = pend.r0.z; sa[3] = pend.r0.w; sa[4] = pend.r1.x; sa[5] = pend.r1.y; sa[6] = pend.r1.z; sa[7] = pend.r1.w;
        }
        if (st && ph >= 1 && ph <= 9) GDT_PIN8(sa);
        if (st && NORM && (ph == 0 || ph == 1)) {
            const float lo = d.in_relu ? 0.f : -3.0e38f;
#pragma unroll
            for (int k = 2 * ph; k < 2 * ph + 2; ++k) {
                const float4 v = nf[k];
                sa[2 * k] = fmaxf(fmaf(sa[2 * k], v.x, v.y), lo);
                sa[2 * k + 1] = fmaxf(fmaf(sa[2 * k + 1], v.z, v.w), lo);
            }
        }
        if (st && NORM && RES && ph == 2) {
            GDT_PIN4(pend.s0); GDT_PIN4(pend.s1);
            sa[0] += pend.s0.x; sa[1] += pend.s0.y; sa[2] += pend.s0.z; sa[3] += pend.s0.w;
            sa[4] += pend.s1.x; sa[5] += pend.s1.y; sa[6] += pend.s1.z; sa[7] += pend.s1.w;
        }
        if (st && WB && ph == 3) *(float4*)((char*)wbf + pend.goff) = make_float4(sa[0], sa[1], sa[2], sa[3]);
        if (st && WB && ph == 4) *(float4*)((char*)wbf + pend.goff + 16) = make_float4(sa[4], sa[5], sa[6], sa[7]);
        if (st && ph == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sa[e] = pend.ok ? sa[e] : 0.f;
            sqlo = 0; sqhi = 0;
        }
#define GDT_Q4P(k)                                                                                                                   \
        if (st && ph == 5 + k) {                                                                                                     \
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(sou[k]) : "v"(sa[2 * k]), "v"(sa[2 * k + 1]));                                  \
            float l0, l1;                                                                                                            \
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(sou[k]), "v"(sa[2 * k]));         \
            asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(sou[k]), "v"(sa[2 * k + 1]));     \
            sqlo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(sqlo, l0, l1, lo_scale, k);                                              \
            sqhi = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(sqhi, __builtin_bit_cast(f16x2, sou[k]), hi_scale, k);                   \
        }
        GDT_Q4P(0) GDT_Q4P(1) GDT_Q4P(2) GDT_Q4P(3)
#undef GDT_Q4P
        if (st && ph == 9) {
            const int row = min((r - SDIST) * RPR + lrow, HROWS_PAD - 1);
            const int phy = (row * 3641) >> 16, phx = row - phy * HW_;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 ov = {sou[0], sou[1], sou[2], sou[3]};
            const int q = lane & 7;
            *(f16x8*)(smem + stage_off + row * ROWB + ((q ^ ((phx >> 1) & 7)) << 4)) = __builtin_bit_cast(f16x8, ov);
            const int key2 = (phx >> 2) & 3;
            const int qo = stage_off + A_BYTES + row * QROWB + ((((q >> 2) << 1) ^ key2) << 4) + ((q & 3) << 2);
            *(unsigned*)(smem + qo) = sqlo;
            *(unsigned*)(smem + (qo ^ 16)) = sqhi;
        }
        if (ld && ph >= 10 && ph <= 13) load_piece_part(ta, chunk, r, ph - 10);
#undef GDT_PIN4
#undef GDT_PIN8
    };

    // ---- weights, streamed L2 -> registers in fragment order (net.hip pack_mx16): per 64 output channels (a wave's slice)
    //   w_c16 [cout/64][K/32][4 blocks][64 lanes][16 B]: lane (n, g) = W[cout block * 16 + n][k = 32 step + 8 g ..+7]
    //   wmx16_a [cout/64][K/64][4][64][16 B] + wmx16_b [..][8 B] + wmx16_s [..][4 B]: lane (n, blk): 32 e2m3 values + E8M0 scale of
    //   blk 0: fp16(w) of k 0-31, 1: w - fp16(w) of k 0-31, 2 / 3: the same of k 32-63 (of the 64 k-values of a (tap, chunk))
    auto wgrp = [&](int tile_n) -> long { return (long)tile_n * 4 + wave; };
    const int nms = d.Kpad >> 6, cin64 = d.Cin >> 6;
    f16x8 bw[RING][4];
    // two sets, used by alternate taps: the set of tap t + 1 is fetched during tap t (its second half-step and its MX run: >= 3000 cycles before the
    // first use).  With one set the loads can only follow the previous tap's MX run, 1000-2000 cycles ahead of their own -- and the weights come
    // from beyond L2 (the activation stream evicts them): every MX run started with a ~440-cycle wait, 16 k cycles per tile (stamped ablations).
    // Tap 0 of a chunk (its set is still in use by tap 8 of the previous chunk: 9 taps, two sets) is fetched in its own first half-step.
    constexpr int BQS = GDT_C16_BQ_SETS;
    v6i bq[BQS + ((GDT_C16_ABL & 64) ? 1 : 0)][4];            // (extra set: ablation 64 only)
    v4i bqs[BQS + ((GDT_C16_ABL & 64) ? 1 : 0)];              // E8M0 scales of the four channel blocks (one dwordx4 per lane)
    auto lane_bytes = [&](int per_lane) -> unsigned {
        unsigned v = lane * per_lane;
        asm volatile("" : "+v"(v));
        return v;
    };
    unsigned lo16 = lane_bytes(16), lo8 = lane_bytes(8), lo4 = lane_bytes(4);
    // ONE record of 15 KB per (64 output channels, 64 k-values): [fp16 step 0: 4 KB][fp16 step 1: 4 KB][MX 16-byte parts: 4 KB][MX 8-byte parts: 2 KB]
    // [scales: 1 KB] -- a wave's whole weight stream is one sequential region
    constexpr long WREC = 15360;
    auto load_bw = [&](int rs, int cb, int tile_n, long ks) {        // ks: uniform index of the 32-k step
        const char* wb = (const char*)d.w_c16 + (wgrp(tile_n) * nms + (ks >> 1)) * WREC + (ks & 1) * 4096;
        bw[rs][cb] = *(const f16x8*)(wb + lo16 + cb * 1024);
    };
    // MX weights of a (tap, chunk): nine loads, ONE per patch row of MFMAs (a vector-memory instruction takes the wave ~16-20 issue cycles, a patch
    // row's four 16-cycle MFMAs cover one of them; three in a row cost 46 cycles each, measured): part 2 cb = the first 16 bytes of block cb's
    // operand tuple, 2 cb + 1 = its last 8 (both loaded INTO the tuple), part 8 = the four blocks' scales
    auto load_bq_part = [&](int set, int part, int tile_n, long ms) {          // ms: uniform index of the 64-k group
        const long f0 = (GDT_C16_ABL & 128) ? 0 : wgrp(tile_n) * nms + ms;      // (ablation 128: every fetch from the same 7 KB)
        const char* rec = (const char*)d.w_c16 + f0 * WREC;
        const int cb = part >> 1;
        if (part == 8) bqs[set] = *(const v4i*)(rec + 14336 + lo16);
        else if ((part & 1) == 0) {
            const v4i qa = *(const v4i*)(rec + 8192 + lo16 + cb * 1024);
            bq[set][cb] = __builtin_shufflevector(__builtin_shufflevector(qa, qa, 0, 1, 2, 3, -1, -1), bq[set][cb], 0, 1, 2, 3, 10, 11);
        } else {
            const v2i qb = *(const v2i*)(rec + 12288 + lo8 + cb * 512);
            bq[set][cb] = __builtin_shufflevector(bq[set][cb], __builtin_shufflevector(qb, qb, 0, 1, -1, -1, -1, -1), 0, 1, 2, 3, 6, 7);
        }
    };

    // ---- activation fragment addresses: lane (n = lane & 15, g = lane >> 4) holds pixel x = PIX(n) of a patch row, k-slot g
    const int fn = lane & 15, fg = lane >> 4;
    const int px = fn < 4 ? 2 * fn : (fn < 12 ? 2 * (fn - 4) + 1 : 2 * (fn - 8));
    int vt[3], vq[3];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
        vt[tx] = px * ROWB + ((fg ^ (((px + tx) >> 1) & 7)) << 4);
        vq[tx] = A_BYTES + px * QROWB + ((fg ^ (((px + tx) >> 2) & 3)) << 4);
    }
    // fp16 fragment of patch row pb, tap (ty, tx), 32-channel half s of the chunk: chunk position (4 s + g) ^ key = ((g ^ key) ^ 4 s)
    auto a_frag = [&](int pb, int ty, int tx, int s) -> f16x8 {
        return *(const f16x8*)(smem + (vt[tx] ^ (s << 6)) + ((pb + ty) * HW_ + tx) * ROWB);
    };
    auto a_qfrag = [&](int pb, int ty, int tx) -> v4i {
        return *(const v4i*)(smem + vq[tx] + ((pb + ty) * HW_ + tx) * QROWB);
    };
    auto flip_stage = [&](int delta) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { vt[k] += delta; vq[k] += delta; }
    };
    // E8M0 scales of the activation side: blocks 0 / 2 carry a_lo (stored * 2^c_lo_exp), 1 / 3 a_hi (stored * 2^-c_hi_exp)
    const int a_scale = (fg & 1) ? 127 + d.c_hi_exp : 127 - d.c_lo_exp;

    const int nchunks = d.Cin >> 6;
    // ---- prologue
#pragma unroll
    for (int u = 0; u < RING - 1; ++u)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) load_bw(u, cb, cur.tile_n, u);
    if (NORM) {
        stage_norm(cur, 0);
        __syncthreads();
    }
    load_nf(0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) store_piece(0, r, load_piece(cur, 0, r));
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SDIST; ++k) pendv[k] = load_piece(cur, 0, 0);      // (placeholder values: overwritten before their first use)

    f16x8 afr[AW];
    v4i aq[QW];
    if (GDT_C16_ABL & 8) { for (int i = 0; i < QW; ++i) aq[i] = a_qfrag(i, 0, 0); }
    if (GDT_C16_ABL & (16 | 64)) { for (int st = 0; st < BQS + ((GDT_C16_ABL & 64) ? 1 : 0); ++st) for (int part = 0; part < 9; ++part) load_bq_part(st, part, cur.tile_n, 0); }
#pragma unroll
    for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0, 0);

    int so = 0;                   // LDS offset of the halo stage of the current chunk (0 or STAGE_BYTES)
    int slot = 0;                 // (scale, shift) slot of the current tile
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
            load_nf(sslot, sc);                 // (the table of the next tile was written during the previous chunk, a barrier ago)
            lo16 = lane_bytes(16); lo8 = lane_bytes(8); lo4 = lane_bytes(4);
            // 32-k step index / tile of half-step u of this chunk; u >= 18: the first half-steps of the chunk staged now (after the very
            // last chunk this fetches the first slices again: unconditional loads keep the code straight-line)
            auto ks_of = [&](int u) -> long { return u < 18 ? (long)(((u >> 1) * cin64 + c) * 2 + (u & 1)) : (long)(sc * 2 + (u - 18)); };
            auto tn_of = [&](int u) -> int { return (u >= 18 && last) ? nxt.tile_n : cur.tile_n; };
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int ty = t / 3, tx = t - ty * 3;
                const int nty = (t + 1) / 3, ntx = (t + 1) - nty * 3;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int u = 2 * t + s;
#pragma unroll
                    for (int pb = 0; pb < 16; ++pb) {
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb)
                            mfma16(acc[pb][cb], bw[u % RING][cb], afr[pb % AW]);     // D[cout][pixel]
                        // the window slot just used takes the fragment AW patch rows on: of this half-step or of the next one
                        if (pb + AW < 16) afr[pb % AW] = a_frag(pb + AW, ty, tx, s);
                        else if (s == 0) afr[pb % AW] = a_frag(pb + AW - 16, ty, tx, 1);
                        else if (t < NTAP - 1) afr[pb % AW] = a_frag(pb + AW - 16, nty, ntx, 0);
                        // weights of half-step u + RING - 1 into the ring slot half-step u - 1 has finished with
                        if (!(GDT_C16_ABL & 4) && (pb & 3) == 2) load_bw((u + RING - 1) % RING, pb >> 2, tn_of(u + RING - 1), ks_of(u + RING - 1));
                        // MX weights of this tap (read by the MX run behind the second half-step; the previous run has finished with them)
                        // MX weights: tap 0's own set in the first nine patch rows of its first half-step; the set of tap t + 1 during the second
                        // half-step of tap t (parts 0-7) and its MX run (the scales)
                        if (!(GDT_C16_ABL & (2 | 16)) && BQS == 2) {
                            if (t == 0 && s == 0 && pb < 9) load_bq_part(0, pb, cur.tile_n, (long)c);
                            if (t + 1 < NTAP && s == 1 && (pb & 1) == 1) load_bq_part((t + 1) & 1, pb >> 1, cur.tile_n, (long)((t + 1) * cin64 + c));
                        }
                        // (one set: behind the previous tap's MX run, one part per second patch row of the first half-step, the scales in the second)
                        if (!(GDT_C16_ABL & (2 | 16)) && BQS == 1 && (pb & 1) == 1 && (s == 0 || pb == 1)) load_bq_part(0, s == 0 ? pb >> 1 : 8, cur.tile_n, (long)(t * cin64 + c));
                        // the first fp4 fragments of the MX run
                        if (!(GDT_C16_ABL & (2 | 8)) && s == 1 && pb >= 16 - QW) aq[pb - (16 - QW)] = a_qfrag(pb - (16 - QW), ty, tx);
                        stage_phase(sta, sc, STAGE_BYTES - so, 3 * t + s, pb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // the correction product of the tap's 64 k-values
#pragma unroll
                for (int pb = 0; pb < 16; ++pb) {
                    if (!(GDT_C16_ABL & 2)) {
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) mfma16_mx(acc[pb][cb], bq[(GDT_C16_ABL & 64) ? BQS : (t % BQS)][cb], aq[pb % QW], bqs[(GDT_C16_ABL & 64) ? BQS : (t % BQS)][cb], a_scale);
                        if (!(GDT_C16_ABL & 8) && pb + QW < 16) aq[pb % QW] = a_qfrag(pb + QW, ty, tx);
                        if (!(GDT_C16_ABL & 16) && BQS == 2 && t + 1 < NTAP && pb == 1) load_bq_part((t + 1) & 1, 8, cur.tile_n, (long)((t + 1) * cin64 + c));
                    }
                    stage_phase(sta, sc, STAGE_BYTES - so, 3 * t + 2, pb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (GDT_C16_ABL & 64) {       // keep the (unused) loads of the two working sets alive
#pragma unroll
                for (int st = 0; st < BQS; ++st) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) asm volatile("" :: "v"(bq[st][cb]));
                    asm volatile("" :: "v"(bqs[st]));
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
                for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0, 0);
            }
        }

        // ------------------------------------------------------------ tile end: all waves are done with the last halo stage and
        // the next tile's first stage (written during the last chunk) is visible
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        GDT_STAMP(st_tbar)
        // the next tile's first fragments are requested before the epilogue: its stores drain under the next tile's first MFMAs
        const int so_next = STAGE_BYTES - so;

        // ------------------------------------------------------------ epilogue, straight from the accumulators (header)
        {
            float* __restrict__ outp = (float*)d.out;
            const float* __restrict__ resp = (const float*)d.res;
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));             // (opaque copy: keeps the epilogue's addresses out of the loop's invariant set)
            const int n_e = lane_e & 15, g_e = lane_e >> 4;
            const int x_e = n_e < 4 ? 2 * n_e : (n_e < 12 ? 2 * (n_e - 4) + 1 : 2 * (n_e - 8));
            const int ch = cur.tile_n * 256 + wave * 64 + 4 * g_e;            // + 16 * channel block
            unsigned o = (unsigned)((cur.n * d.H + cur.y0) * d.W + cur.x0 + x_e) * (unsigned)d.Cout + (unsigned)ch;
            const unsigned rowstep = (unsigned)d.W * (unsigned)d.Cout;
            const bool lowhalf = n_e < 8;
            const int n_p = n_e ^ 8, x_p = n_p < 4 ? 2 * n_p : (n_p < 12 ? 2 * (n_p - 4) + 1 : 2 * (n_p - 8));      // the partner lane's pixel
            const unsigned o_p = o + (unsigned)((x_p - x_e) * d.Cout);
            unsigned oA = (lowhalf ? o : o_p) + (lowhalf ? 0u : 16u), oB = (lowhalf ? o_p : o) + (lowhalf ? 0u : 16u);
            const float lo = d.relu ? 0.f : -__builtin_inff();
            constexpr bool EPI_RES = MODE == 0;
            const bool has_res = EPI_RES && resp != nullptr;
            float4 bv[4], s1[4], s2[4];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                bv[cb] = d.bias ? *(const float4*)(d.bias + ch + cb * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                s1[cb] = make_float4(0.f, 0.f, 0.f, 0.f); s2[cb] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4 rcur[4], rnxt[4];
            if (has_res) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) rcur[cb] = *(const float4*)(resp + o + cb * 16);
            }
#pragma unroll
            for (int pb = 0; pb < 16; ++pb) {
                asm volatile("" : "+v"(o));
                if (has_res && pb + 1 < 16) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) rnxt[cb] = *(const float4*)(resp + o + rowstep + cb * 16);
                }
                float4 tv[4];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    const f32x4& a = acc[pb][cb];
                    float4 t = make_float4(a[0] + bv[cb].x, a[1] + bv[cb].y, a[2] + bv[cb].z, a[3] + bv[cb].w);
                    s1[cb].x += t.x; s1[cb].y += t.y; s1[cb].z += t.z; s1[cb].w += t.w;
                    s2[cb].x += t.x * t.x; s2[cb].y += t.y * t.y; s2[cb].z += t.z * t.z; s2[cb].w += t.w * t.w;
                    if (has_res) { t.x += rcur[cb].x; t.y += rcur[cb].y; t.z += rcur[cb].z; t.w += rcur[cb].w; }
                    t.x = fmaxf(t.x, lo); t.y = fmaxf(t.y, lo); t.z = fmaxf(t.z, lo); t.w = fmaxf(t.w, lo);
                    tv[cb] = t;
                }
                if (GDT_C16_ABL & 256) {      // (ablation 256: no output stores)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) asm volatile("" :: "v"(tv[cb].x), "v"(tv[cb].y), "v"(tv[cb].z), "v"(tv[cb].w));
                } else if (!GDT_C16_FULL_LINES) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) *(float4*)(outp + o + cb * 16) = tv[cb];      // 16 pixels x 64 bytes per instruction
                } else {
                    // Whole lines: as the registers stand an instruction writes 64 bytes (the four lane groups) of each of its 16 pixels.  The pixel
                    // lanes n and n ^ 8 swap one block of a pair (2p, 2p + 1): lanes n < 8 hand over block 2p + 1 and receive block 2p of the partner,
                    // so that instruction A writes blocks 2p | 2p + 1 = 128 contiguous bytes of the pixels on lanes 0-7, B those of lanes 8-15.
                    auto ror8 = [](float v) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)); };
#pragma unroll
                    for (int p2 = 0; p2 < 2; ++p2) {
                        const float4 t0 = tv[2 * p2], t1 = tv[2 * p2 + 1];
                        // (component-wise selects: a ?: on the float4 structs becomes an indexed scratch array)
                        const float rx = ror8(lowhalf ? t1.x : t0.x), ry = ror8(lowhalf ? t1.y : t0.y), rz = ror8(lowhalf ? t1.z : t0.z), rw = ror8(lowhalf ? t1.w : t0.w);
                        *(float4*)(outp + oA + p2 * 32) = make_float4(lowhalf ? t0.x : rx, lowhalf ? t0.y : ry, lowhalf ? t0.z : rz, lowhalf ? t0.w : rw);
                        *(float4*)(outp + oB + p2 * 32) = make_float4(lowhalf ? rx : t1.x, lowhalf ? ry : t1.y, lowhalf ? rz : t1.z, lowhalf ? rw : t1.w);
                    }
                    oA += rowstep; oB += rowstep;
                }
                o += rowstep;
                if (has_res) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) rcur[cb] = rnxt[cb];
                }
            }
            if (d.stats) {
                // sum over the 16 pixel lanes of a lane group (a DPP row): rotate butterfly, every lane ends with the total, fixed order
                auto merge = [](float v) -> float {
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));    // row_ror:8
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));    // row_ror:4
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));    // row_ror:2
                    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));    // row_ror:1
                    return v;
                };
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    s1[cb].x = merge(s1[cb].x); s1[cb].y = merge(s1[cb].y); s1[cb].z = merge(s1[cb].z); s1[cb].w = merge(s1[cb].w);
                    s2[cb].x = merge(s2[cb].x); s2[cb].y = merge(s2[cb].y); s2[cb].z = merge(s2[cb].z); s2[cb].w = merge(s2[cb].w);
                }
                if (n_e == 0) {
                    // slab layout of the 256-row patch kernels: two 128-row records per patch; this wave's 256 rows go into the first one
                    float* dst = d.stats + ((long)(d.stats_tile_base + cur.tile_m * 2) * 2) * d.Cout + ch;
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        *(float4*)(dst + cb * 16) = s1[cb];
                        *(float4*)(dst + d.Cout + cb * 16) = s2[cb];
                        *(float4*)(dst + 2l * d.Cout + cb * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
                        *(float4*)(dst + 3l * d.Cout + cb * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            }
        }
#ifdef GDT_C_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (diagnostic only: the epilogue's stores are charged to the epilogue)
        GDT_STAMP(st_epi)
        ++st_n;
#endif
        if (!nxt.valid) break;
        cur = nxt; vb += gridDim.x; slot ^= 1;
        flip_stage(so_next - so);
        so = so_next;
#pragma unroll
        for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0, 0);
    }
#ifdef GDT_C_STAMP
    if (lane == 0 && d.stamp_out) {
        unsigned long long* o = d.stamp_out + ((long)blockIdx.x * 4 + wave) * 8;
        o[0] = st_body; o[1] = st_cbar; o[2] = st_tbar; o[3] = st_epi; o[4] = __builtin_amdgcn_s_memtime() - st_begin; o[5] = st_n;
    }
#endif
}

template <int MODE>
int launch_c16(const ConvLaunch& d, hipStream_t stream) {
    const int tiles = d.N * (d.W >> 4) * (d.H >> 4), ntn = d.CoutPad >> 8;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_c16_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
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
    hipLaunchKernelGGL((conv3x3_halo_c16_kernel<MODE>), dim3(grid), dim3(NT), LDS_BYTES, stream, ds, vblocks);
    if (++stamp_calls % 200 < 20) {
        GDT_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)grid * 4 * 8);
        GDT_CHECK_HIP(hipMemcpy(h.data(), stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s[6] = {0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)grid * 4; ++w) for (int k = 0; k < 6; ++k) s[k] += (double)h[w * 8 + k];
        const double nw = grid * 4.0, nt = s[5] / nw;
        fprintf(stderr, "[c stamp] MODE %d FORM 16 BN 256 waves 4: tiles/wave %.1f; per tile: chunk bodies %.0f, chunk barriers %.0f, tile barrier %.0f, epilogue %.0f cycles; total per wave %.0f\n",
                MODE, nt, s[0] / nw / nt, s[1] / nw / nt, s[2] / nw / nt, s[3] / nw / nt, s[4] / nw);
    }
#else
    hipLaunchKernelGGL((conv3x3_halo_c16_kernel<MODE>), dim3(grid), dim3(NT), LDS_BYTES, stream, d, vblocks);
#endif
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// Eligible: what conv3x3_halo_c.hip's 256-column form takes (gdt_conv_halo_c_eligible is checked by the caller), restricted to whole 16 x 16
// patches and whole 256-column tiles, with the 16 x 16 fragment-ordered weights present.
bool gdt_conv_halo_c16_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_HALO_C16"); return e ? atoi(e) : 1; }();   // 0 off
    if (mode == 0 || !d.w_c16) return false;
    if (!gdt_conv_halo_c_eligible(d) || gdt_conv_halo_c_columns(d) != 256) return false;
    if ((d.H & 15) || (d.W & 15) || d.Cout != d.CoutPad || (d.Cout & 255)) return false;
    if (d.res && d.in_norm) return false;
    return true;
}

int gdt_launch_conv_halo_c16(const ConvLaunch& d, hipStream_t stream) {
    if (!d.in_norm) return launch_c16<0>(d, stream);
    if (d.in_res) return d.in_out ? launch_c16<7>(d, stream) : launch_c16<3>(d, stream);
    return d.in_out ? launch_c16<5>(d, stream) : launch_c16<1>(d, stream);
}
