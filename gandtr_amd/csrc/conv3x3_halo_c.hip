// 3x3 / stride-1 / pad-1 convolution (and ConvTranspose2d(k3,s2,p1,op1), CT form) in the "f16c" precision mode: fp32 NHWC
// activations, fp16 MFMA main product plus a block-scaled (MX) low-precision product that carries both rounding residuals.
//
// Why: single-pass fp16 (11-bit operands) leaves the 24-layer random-weight generator at 2e-3 of the fp32 reference, above
// north_star's 1e-3; the exact split (f16x3: a_hi*w_hi + a_lo*w_hi + a_hi*w_lo, three fp16 passes) costs 3x the matrix work.
// The two correction terms are ~2^-11 of the result, so 2-4 significant bits are enough for them:
//     a*w  ~=  a_hi*w_hi  (v_mfma_f32_32x32x16_f16, exact products, fp32 accumulate)
//           +  [a_lo | a_hi]_fp4 . [w_hi | w_lo]_fp6   (ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32 k-values: its two 32-wide
//              K blocks hold the two correction terms; per-block E8M0 scales undo the 2^12 / per-block weight scaling)
// = 1.5x the fp16 work at the fp4/fp6 MFMA rate (measured 1.43x, profiles/experiments/mx_probe.hip) instead of 3x.  Measured
// against the fp32 oracle: <= 4e-4 at every tap of the full-size generator (single-pass fp16: 2.5e-3; f16x3: 3e-6).
//
// Structure (the shipped forms; conv3x3_halo_c16.hip is the same layer on the 16 x 16 MFMA shapes and takes the 256-column resblock convs of whole
// 16 x 16 patches since round 4 -- this file keeps the 128-column form for few patches, ragged image sizes, and the shift forms):
//   * persistent workgroups walking an XCD-chunked tile list; a tile = a 16 x 16 output patch x 256 (or 128) output channels; swapped MFMA
//     operands, D[cout][pixel] = W . A^T;
//   * FOUR waves, one per SIMD, 512 registers each (256 accumulator AGPRs + 256 VGPRs), laid out 1 x 4 in the 3x3 and stride-2 forms: every wave
//     owns all 256 pixels x 64 output channels, so a weight fragment feeds 8 MFMAs (half the L1 weight bytes per MFMA of a 2 x 2 layout, twice its LDS
//     fragment bytes); the transposed form and the 128-column tiles keep 2 x 2 waves (its phase-per-block map needs four blocks per wave); an
//     experimental 1 x 8 layout -- two waves per SIMD at 256 registers, each 256 pixels x 32 channels -- exists behind GDT_C_WAVES=8 (same cycles,
//     lower clock: off);
//   * the halo (18 x 18 per 64-channel chunk; 17 x 17 in the shift forms) is read as fp32 through registers, branch-free, one loader round in flight per
//     thread, normalised / ReLU'd / residual-added in fp32 (the producer's InstanceNorm folded in: MODE bits 1 norm, 2 + residual, 4 + write-back of
//     the transformed tensor), then split by v_cvt_pk_f16_f32 / v_fma_mix_f32 / v_cvt_scalef32_pk_fp4_*: fp16 hi plane (128-byte rows, XOR swizzle) and
//     an fp4 plane (64-byte rows: [lo 0-31][hi 0-31][lo 32-63][hi 32-63], own swizzle), two stages in LDS, one barrier per chunk;
//   * weights stream L2 -> registers in MFMA fragment order: fp16 fragments in a ring of 3-4 substep slices, MX fragments (a lane's 32 e2m3 values = 16 + 8
//     bytes + an E8M0 scale dword, three contiguous arrays) loaded INTO the MFMA's 6-register operand tuple; scalar-base addressing;
//   * per k-substep (16 k-values): 16 fp16 MFMAs per wave, every second substep 16 MX MFMAs on the 32 k-values just done; MFMAs row-outer, the
//     activation fragment re-loaded in place; sched_group_barrier lays the substep out as MFMA, one LDS read, <= 4 VALU, a memory operation per two MFMAs;
//   * epilogue, wave-private and software-pipelined: every 32 x 32 accumulator block goes through a 4 KB XOR-swizzled patch of the wave's own (two
//     patches: block n + 1 is written while block n's four row reads are in flight) and leaves as whole 128-byte lines; interior tiles run a
//     branch-free body (running per-lane store offset, ReLU as a max with 0 / -inf, the epilogue residual of the BatchNorm generator fetched one
//     block ahead); InstanceNorm statistics: ONE 256-row record per wave (the second 128-row record of the slab layout is written as zeros), the
//     8 pixel lanes merged by a DPP rotate + a swizzle + one permute per value, fixed order => deterministic;
//   * FORM 1 (transposed): a wave's four 32-column blocks are the four sub-pixel phases of 32 output channels, so the 7 zero (shift, phase) blocks of 16
//     are skipped at compile time; FORM 2 (stride 2): the loader reads the virtual space-to-depth view of its input, 9 of the 16 (shift, parity) blocks
//     are non-zero (skipped per chunk: compile-time masks in the 128-column form, run-time in the 256-column one).
#include <cstdio>
#include <cstdlib>

#include <type_traits>
#include <vector>

#include "gdt_common.h"

// timing-only ablations (profiles/experiments): compile with -DGDT_C_ABL=<bits>; results are wrong by design
//   1 no halo staging after the prologue   2 no MX MFMAs / MX weight loads   4 no fp16 weight re-loads   8 no fp16 MFMAs   256 no fp16 fragment re-loads
#ifndef GDT_C_ABL
#define GDT_C_ABL 0
#endif
// transposed form: skip the all-zero (shift, phase) weight blocks under a run-time (wave-uniform) mask?  Measured: the branches around
// the MFMAs of a one-wave-per-SIMD loop cost ~190 spilled registers; issuing the zero blocks (16/9 of the MFMAs) is faster.
#ifndef GDT_C_DEPTH
#define GDT_C_DEPTH 1           // loader rounds in flight per thread, 3x3 form (modes without / with residual + write-back)
#endif
#ifndef GDT_C_DEPTH_RES
#define GDT_C_DEPTH_RES 1
#endif
#ifndef GDT_C_DEPTH_SHIFT
#define GDT_C_DEPTH_SHIFT 2     // ... shift forms (stride-2, transposed): little matrix work per halo byte, the rounds in flight set the HBM rate
#endif
#ifndef GDT_C_DEPTH_S2_W8
#define GDT_C_DEPTH_S2_W8 2     // ... the eight-wave 128-column stride-2 form (registers to spare)
#endif
#ifndef GDT_C_DEPTH_SHIFT_RES
#define GDT_C_DEPTH_SHIFT_RES 2
#endif
// cache policy of the streamed loads: bit 1 = weight fragments, bit 2 = halo pieces are fetched non-temporal (the line is not kept in the
// CU's 32 KB L1: with two substeps of weights per wave in flight the outstanding lines alone fill it)
#ifndef GDT_C_NT
#define GDT_C_NT 0
#endif
#ifndef GDT_C_WB_INTERIOR
#define GDT_C_WB_INTERIOR 0     // 1: write-back stores of the folded norm only for the patch's interior pixels (exec-masked stores)
#endif
#ifndef GDT_C_SCHED
#define GDT_C_SCHED 2          // 1: loads after each column's MFMAs (clumped)   2: one load per two MFMAs (+0.3 % images/s)
#endif
#ifndef GDT_C_WSTAG
#define GDT_C_WSTAG 0           // s_sleep units (64 cycles) of start offset between consecutive waves after each chunk barrier
#endif
#ifndef GDT_C_CT_SKIP
#define GDT_C_CT_SKIP 1
#endif

namespace {

constexpr int ROWB = 128;          // bytes per row of the fp16 plane (64 halves of K)
constexpr int QROWB = 64;          // bytes per row of the fp4 plane
constexpr int HALO_W = 18;
constexpr int PH = 16;
constexpr int HALO_ROWS_PAD = 328;
constexpr int A_BYTES = HALO_ROWS_PAD * ROWB;              // 41984
constexpr int Q_BYTES = HALO_ROWS_PAD * QROWB;             // 20992
constexpr int STAGE_BYTES = A_BYTES + Q_BYTES;             // 62976
constexpr int NORM_BYTES = 4096 + 64;                      // (scale, shift): two slots of up to 256 input channels + a zero entry
constexpr int BM = PH * 16;
// + epilogue patches: two 4 KB patches per wave with four waves (pipelined body), one with eight (all 160 KB are taken then)
constexpr size_t lds_bytes(int waves) { return 2 * (size_t)STAGE_BYTES + NORM_BYTES + (waves > 4 ? waves * 4096 : 4 * 8192); }

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

struct TileAt { int n, y0, x0, tile_m, tile_n; bool valid; };

// MODE bits: 1 = the producer's InstanceNorm (+ReLU) is applied while staging; 2 = ... plus a residual; 4 = the transformed
// tensor is written back.  CT: transposed form (see conv3x3_halo_rb.hip).
// FORM 0: 3x3 / stride 1.  FORM 1 (CT): ConvTranspose2d(k3,s2,p1,op1), see conv3x3_halo_rb.hip.  FORM 2 (S2): Conv2d(k3, s2, p1, zero
// padding) as a 2x2-shift convolution over the space-to-depth view of its input (x[2R+py][2C+px][c] = channel (2py+px)*Cin + c of
// S2D pixel (R, C)): out[y][x] = sum over shifts (dy, dx) in {-1, 0}^2 and the 4*Cin virtual channels, with the (shift, parity)
// pairs that do not occur as zero weight blocks that are skipped (9 of 16 remain).  The S2D tensor is never materialised: the halo
// loader maps (S2D pixel, virtual channel) to the NHWC address.  16x16 OUTPUT patch, 17x17 S2D halo, the 4 shifts as taps.
template <int BN, int WGM, int WGN, int MODE, int FORM = 0>
__global__ __launch_bounds__(WGM * WGN * 64) void conv3x3_halo_c_kernel(const ConvLaunch d, const int vblocks) {
    constexpr bool CT = FORM == 1, S2 = FORM == 2, SHIFT = FORM != 0;
    constexpr bool NORM = (MODE & 1) != 0, RES = (MODE & 2) != 0, WB = (MODE & 4) != 0;
    constexpr int NT = WGM * WGN * 64, RPR = NT / 8;   // threads, halo rows staged per loader round
    constexpr int HW_ = SHIFT ? 17 : HALO_W;
    constexpr int HROWS = HW_ * HW_, HROWS_PAD = (HROWS + 7) / 8 * 8;
    constexpr int NTAP = SHIFT ? 4 : 9;
    constexpr int NR = (HROWS_PAD + RPR - 1) / RPR;
    // staging schedule: a chunk has NTAP * 4 k-substep slots; loader round r is issued at slot r * SPR and written to LDS at slot
    // (r + 1) * SPR (one piece in flight per thread, SPR substeps of MFMAs to cover its latency)
    // (the shift forms have 16 substeps for 10 rounds, one substep apart: they keep DEPTH = 2 rounds in flight instead)
    constexpr int SLOTS = NTAP * 4, DEPTH = (FORM == 2 && WGM * WGN == 8) ? GDT_C_DEPTH_S2_W8 : SHIFT ? (RES ? GDT_C_DEPTH_SHIFT_RES : GDT_C_DEPTH_SHIFT) : ((MODE & 6) ? GDT_C_DEPTH_RES : GDT_C_DEPTH), SPR = SLOTS / (NR + DEPTH);
    static_assert(SPR >= 1 && (NR + DEPTH - 1) * SPR < SLOTS && HROWS_PAD <= HALO_ROWS_PAD, "halo rounds are spread over the substeps of the previous chunk");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    static_assert((TM == 4 || TM == 8) && (TN == 1 || TN == 2 || TN == 4) && TM * TN <= 16, "tile shape (128-row statistics records: one or two per wave)");
    constexpr int NWAVES = WGM * WGN;
    constexpr bool PIPE2 = NWAVES <= 4;             // two epilogue patches per wave
    // TN == 1 (eight waves of 256 pixels x 32 channels, two per SIMD at 256 registers): an activation fragment feeds ONE MFMA, so the
    // fragments rotate through a window of AW registers sets, each re-loaded AW MFMAs ahead of its use (TN >= 2: AW = TM, the fragment
    // of row block i is re-loaded in place for the next substep)
    constexpr int AW = TN == 1 ? 4 : TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Eight waves at 256 registers: nothing that only depends on the lane id may stay live across the chunk loop (the allocator spills it and every
    // scratch reload is a vector-memory wait behind the halo stream) -- such values are re-made from v_mbcnt where they are used.
    constexpr bool REMAT = WGM * WGN > 4;
    auto lane_now = [&]() -> int {
        if (!REMAT) return lane;
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto tid_now = [&]() -> int { return REMAT ? wave * 64 + lane_now() : tid; };
    const int wm = wave / WGN, wn = wave % WGN;
    const float* __restrict__ inf = (const float*)d.in;
    const float* __restrict__ resf = (const float*)d.in_res;
    float* __restrict__ wbf = (float*)d.in_out;

    const int GH = S2 ? d.OH : d.H, GW = S2 ? d.OW : d.W;              // the grid the 16x16 patches tile (S2: the output)
    const int tiles_x = (GW + 15) >> 4, tiles_y = (GH + PH - 1) / PH;
    const int tpi = tiles_x * tiles_y, ntm = d.N * tpi, ntn = d.CoutPad / BN;
    auto tile_at = [&](int vb) -> TileAt {
        TileAt t;
        t.valid = vb < vblocks && gdt_tile_of_block(vb, ntm, ntn, t.tile_m, t.tile_n);
        if (!t.valid) { t.tile_m = 0; t.tile_n = 0; }
        t.n = t.tile_m / tpi;
        const int tr = t.tile_m - t.n * tpi;
        t.y0 = (tr / tiles_x) * PH; t.x0 = (tr % tiles_x) << 4;
        return t;
    };
    int vb = blockIdx.x;
    TileAt cur = tile_at(vb);
    if (!cur.valid) return;
    // Stagger: every workgroup runs the same number of equally long tiles, so without it all 256 CUs reach their epilogue in the
    // same microsecond, the chip's whole output (64 MB per round at batch 64) leaves in one burst and every wave then sits behind its
    // own stores (in-order vmcnt) for the ~20 us the burst takes to drain.  Four phase groups, d.stagger_us apart, spread it.
    if (d.stagger_us > 0) {
        const int grp = (blockIdx.x >> 3) & 3;
        const unsigned long long t0 = __builtin_readcyclecounter();
        const unsigned long long wait = (unsigned long long)grp * d.stagger_us * 2000;      // s_memtime counts shader cycles (~2 GHz under load)
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }

    // ---- halo loader: through registers, branch-free (see conv3x3_halo_rb.hip for why)
    const int lrow = tid >> 3;
    const bool refl = d.pad_reflect != 0;
    const float lo_scale = __builtin_ldexpf(1.f, -d.c_lo_exp), hi_scale = __builtin_ldexpf(1.f, d.c_hi_exp);   // cvt divides by the scale
    struct Pend { float4 r0, r1, s0, s1; unsigned goff; bool ok; };
    auto load_piece = [&](const TileAt& ta, int chunk, int r) -> Pend {
        const int tl = tid_now();
        int lr = REMAT ? tl >> 3 : lrow;
        asm volatile("" : "+v"(lr));
        const int h = min(r * RPR + lr, HROWS_PAD - 1);
        const int hy = (h * (SHIFT ? 3856 : 3641)) >> 16, hx = h - hy * HW_;
        const int q = tl & 7;          // a lane always fetches the same 8-channel group of its pixel; the swizzle is applied at the LDS write
        int iy = ta.y0 - (CT ? 0 : 1) + hy, ix = ta.x0 - (CT ? 0 : 1) + hx;
        int cbyte = (chunk * 8 + q) * 32;                            // byte offset of this piece's 8 channels within its pixel
        if (S2) {                                                     // virtual channel -> (sub-pixel parity, real channel)
            const int vch = chunk * 64 + q * 8, lcr = d.lc8 + 1;      // cr = Cin / 4 real channels = 2^lcr (a power of two >= 64:
            const int par = vch >> lcr;                               //  a chunk never straddles parities; shifts, not a division)
            iy = 2 * iy + (par >> 1); ix = 2 * ix + (par & 1);
            cbyte = (vch & ((1 << lcr) - 1)) * 4;
        }
        int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
        int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
        ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);
        const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
        Pend p;
        p.goff = (((unsigned)((ta.n * d.H + ry) * d.W + rx) << (d.lc8 + (S2 ? 3 : 5))) + cbyte);      // byte offset (< 2^32, checked on the host)
        p.ok = (h < HROWS) & (inb | refl);
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        auto ldh = [](const float4* q) -> float4 {
            if constexpr ((GDT_C_NT & 2) != 0) { const f32x4 v = __builtin_nontemporal_load((const f32x4*)q); return make_float4(v[0], v[1], v[2], v[3]); }
            else return *q;
        };
        p.r0 = ldh((const float4*)((const char*)inf + p.goff)); p.r1 = ldh((const float4*)((const char*)inf + p.goff + 16));
        if (RES && !(GDT_C_ABL & 32)) { p.s0 = ldh((const float4*)((const char*)resf + p.goff)); p.s1 = ldh((const float4*)((const char*)resf + p.goff + 16)); }
        return p;
    };
    float* nlds = (float*)(smem + 2 * STAGE_BYTES);
    auto stage_norm = [&](const TileAt& ta, int slot) {
        const int creal = S2 ? d.Cin >> 2 : d.Cin;
        for (int i = tid_now(); i < creal / 2; i += NT) {              // float4 = 2 channels x (mean, rstd) -> (scale, shift)
            const float4 v = *(const float4*)(d.in_norm + (long)ta.n * creal * 2 + i * 4);
            *(float4*)(nlds + slot * 512 + i * 4) = make_float4(v.y, -v.x * v.y, v.w, -v.z * v.w);
        }
    };
    // (scale, shift) of the lane's 8 channels in the chunk being staged: a lane's channel group is the same for every piece of a chunk,
    // so the factors are fetched from the LDS table once per chunk, not per piece (per piece: four dependent LDS reads, each behind an
    // `s_waitcnt lgkmcnt(0)` that also drained the fragment reads in flight)
    // (eight waves, 256 registers each: the 16 factor registers are not kept across the chunk -- a piece reads them at its store: six
    // pieces per chunk and thread, and the partner wave of the SIMD covers the LDS latency)
    constexpr bool NF_REGS = NWAVES <= 4;
    float4 nf[4];
    const float4* nf_ptr = nullptr;
    auto load_nf = [&](int slot, int chunk) {
        if (!NORM) return;
        const int l7 = lane_now() & 7;
        const int cq = S2 ? ((chunk * 64 + l7 * 8) & ((1 << (d.lc8 + 1)) - 1)) >> 3 : chunk * 8 + l7;    // the (real) 8-channel group
        const float4* np4 = (const float4*)(nlds + slot * 512 + cq * 16);
        if (NF_REGS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) nf[k] = np4[k];
        } else nf_ptr = np4;
    };
    auto store_piece = [&](int stage_off, int r, const Pend& p) {
        const int tl = tid_now();
        const int row = min(r * RPR + (REMAT ? tl >> 3 : lrow), HROWS_PAD - 1);
        const int phy = (row * (SHIFT ? 3856 : 3641)) >> 16, phx = row - phy * HW_;
        float a[8] = {p.r0.x, p.r0.y, p.r0.z, p.r0.w, p.r1.x, p.r1.y, p.r1.z, p.r1.w};
        if (NORM) {
            const float lo = d.in_relu ? 0.f : -3.0e38f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = NF_REGS ? nf[k] : nf_ptr[k];
                a[2 * k] = fmaxf(fmaf(a[2 * k], v.x, v.y), lo);
                a[2 * k + 1] = fmaxf(fmaf(a[2 * k + 1], v.z, v.w), lo);
            }
            if (RES) {
                a[0] += p.s0.x; a[1] += p.s0.y; a[2] += p.s0.z; a[3] += p.s0.w;
                a[4] += p.s1.x; a[5] += p.s1.y; a[6] += p.s1.z; a[7] += p.s1.w;
            }
        }
        // write-back of the transformed tensor (every piece stores the value of its clamped source pixel: identical bits from
        // neighbouring patches, no branch)
        if (WB && !(GDT_C_ABL & 16)) {
#if GDT_C_WB_INTERIOR
            // only the patch's own 16 x 16 pixels: the halo ring belongs to the neighbours, who write identical bits (27 % of the stores)
            if ((unsigned)(phy - 1) < 16u && (unsigned)(phx - 1) < 16u) {
                *(float4*)((char*)wbf + p.goff) = make_float4(a[0], a[1], a[2], a[3]);
                *(float4*)((char*)wbf + p.goff + 16) = make_float4(a[4], a[5], a[6], a[7]);
            }
#else
            *(float4*)((char*)wbf + p.goff) = make_float4(a[0], a[1], a[2], a[3]);
            *(float4*)((char*)wbf + p.goff + 16) = make_float4(a[4], a[5], a[6], a[7]);
#endif
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = p.ok ? a[e] : 0.f;
        // split: o = fp16(a) (round to nearest even, two per instruction), a_lo = a - o in ONE v_fma_mix_f32 each (fp16 source read
        // in place), both planes quantised to fp4 by the scaled converts (the convert divides by its scale operand)
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
        const f16x8 o = __builtin_bit_cast(f16x8, ov);
        *(f16x8*)(smem + stage_off + row * ROWB + (((tl & 7) ^ ((phx >> 1) & 7)) << 4)) = o;      // source chunk q at position q ^ key (a_frag)
        // fp4 plane: this thread holds source chunk q (channels 8q .. 8q+7): 32-channel block b = q >> 2, dword q & 3;
        // 16-byte position (2b + {lo 0, hi 1}) ^ key, key = conflict-free swizzle of the fragment reads (see a_qfrag)
        const int q = tl & 7;
        const int key = SHIFT ? ((phy + 2 * (phx >> 2)) & 3) : ((phx >> 1) & 3);
        const int qo = stage_off + A_BYTES + row * QROWB + ((((q >> 2) << 1) ^ key) << 4) + ((q & 3) << 2);
        *(unsigned*)(smem + qo) = qlo;
        *(unsigned*)(smem + (qo ^ 16)) = qhi;
    };

    // ---- weights, streamed L2 -> registers in MFMA fragment order.  Grouped layouts (net.hip pack_mx): the four 32-channel
    // fragments a wave needs for one k-substep are contiguous, so one scalar base + 32-bit lane offset + an immediate (j * 1 KB)
    // addresses each of them (the scalar-base form of global_load: no 64-bit vector address arithmetic):
    //   w_cfrag [cout/128][K/16][4][64 lanes][16 B]    wmx_a [cout/128][K/32][4][64][16 B]    wmx_b [..][64][8 B]    wmx_s [..][64][4 B]
    // (a lane's 32 e2m3 values = 16 bytes of wmx_a + 8 of wmx_b, fetched by a dwordx4 and a dwordx2 load INTO the MFMA's 6-register
    // operand tuple; its E8M0 block scale = a dword of wmx_s.  Three arrays because (a) any load whose register tuple is not exactly a
    // sub-tuple of that operand -- the earlier 16 + 12-byte split with the scale behind the values, or two dwordx4 into an 8-tuple --
    // makes the compiler land it in scratch registers and copy it over behind an `s_waitcnt vmcnt(0..4)` at the top of the NEXT
    // substep: one exposed L2 latency per 32 k-values; (b) adjacent loads from one base are merged back into such a load by the
    // load vectoriser; (c) every array must stay contiguous over the 64 lanes of a fragment: 32-byte per-lane records made each of the
    // three loads touch 16 cache lines instead of 8 + 4 + 2 and the L1 line rate, not the bytes, is what the weight stream costs.)
    // One wave per SIMD issues in order, so no load may wait on the MFMAs of its own substep: the fp16 fragments live in a ring of
    // RING substep slices and substep u re-loads the slot that substep u - 1 has just finished with (slice u + RING - 1); the MX
    // fragments of group g (32 k-values, MFMAs at the end of substep 2g + 1) are re-loaded with group g + 1 early in substep 2g + 2.
    // Each substep issues its loads row by row between its MFMAs.
#ifndef GDT_C_RING
#define GDT_C_RING 3
#endif
    constexpr int RING = SHIFT ? 4 : GDT_C_RING;
    static_assert(SLOTS % RING == 0, "ring / buffer positions of a substep must not depend on the chunk");
    static_assert(TN == 4 || TN == 2 || TN == 1, "the weight streams are grouped per 128 output channels");
    const int wgrp_of_wave = (wn * WTN) >> 7, wblk = ((wn * WTN) >> 5) & 3;      // the wave's 128-column weight group within the tile, its first 32-column block in it
    const int nks = d.Kpad >> 4, nms = d.Kpad >> 5, cin16 = d.Cin >> 4;
    f16x8 b[RING][TN];
    typedef int v6i __attribute__((ext_vector_type(6)));
    v6i bq[TN];             // the lane's 32 e2m3 values: the MFMA's 6-register operand tuple, loaded in place (dwordx4 + dwordx2)
    int bqs[TN];            // its E8M0 block scale
    auto lane_bytes = [&](int per_lane) -> unsigned {     // (opaque: keeps the zero-extension next to its load, which is what lets the
        unsigned v = lane_now() * per_lane;                //  compiler pick the scalar-base addressing form)
        asm volatile("" : "+v"(v));
        return v;
    };
    unsigned lo16 = lane_bytes(16), lo8 = lane_bytes(8), lo4 = lane_bytes(4);        // (refreshed at the top of every chunk body)
    auto ldw = [](const auto* p) { if constexpr ((GDT_C_NT & 1) != 0) return __builtin_nontemporal_load(p); else return *p; };
    auto load_b = [&](int rs, int j, int tile_n, long ks) {       // ks: uniform k-substep index (16 k-values each)
        const char* wb = (const char*)d.w_cfrag + ((long)(tile_n * (BN / 128) + wgrp_of_wave) * nks + ks) * 4096;
        b[rs][j] = ldw((const f16x8*)(wb + lo16 + (wblk + j) * 1024));
    };
    auto load_bq = [&](int j, int tile_n, long ks) {     // MX fragment j of the 32-k group starting at substep ks (even)
        const long f0 = (long)(tile_n * (BN / 128) + wgrp_of_wave) * nms + (ks >> 1);
        const v4i qa = ldw((const v4i*)((const char*)d.wmx_a + f0 * 4096 + lo16 + (wblk + j) * 1024));
        const v2i qb = ldw((const v2i*)((const char*)d.wmx_b + f0 * 2048 + lo8 + (wblk + j) * 512));
        bq[j] = __builtin_shufflevector(__builtin_shufflevector(qa, qa, 0, 1, 2, 3, -1, -1), __builtin_shufflevector(qb, qb, 0, 1, -1, -1, -1, -1), 0, 1, 2, 3, 6, 7);
        bqs[j] = ldw((const int*)((const char*)d.wmx_s + f0 * 1024 + lo4 + (wblk + j) * 256));
    };

    // A fragment addresses (see conv3x3_halo_rb.hip); fp4 plane: per-lane base per tap column (+ tap row for CT) with the swizzle key
    const int fr = lane & 31, fh = lane >> 5;
    int vt[3], vq[SHIFT ? 4 : 3];
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
        vt[tx] = ((wm * (WTM / 16) + (fr >> 4)) * HW_ + (fr & 15)) * ROWB + ((fh ^ ((((fr & 15) + tx) >> 1) & 7)) << 4);
    if (!SHIFT) {
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
            vq[tx] = A_BYTES + ((wm * (WTM / 16) + (fr >> 4)) * HW_ + (fr & 15)) * QROWB + ((fh ^ ((((fr & 15) + tx) >> 1) & 3)) << 4);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {       // key = (hy + 2 * (hx >> 2)) & 3, hy = wm*8 + (fr>>4) + 2i + ty: the 2i term is XORed in at the read
            const int ty = t >> 1, tx = t & 1;
            const int key = (((fr >> 4) + ty) + 2 * ((((fr & 15) + tx) >> 2) & 1)) & 3;
            vq[t] = A_BYTES + ((wm * (WTM / 16) + (fr >> 4)) * HW_ + (fr & 15)) * QROWB + ((fh ^ key) << 4);
        }
    }
    // (the stage offset `so` is folded into vt / vq when the stage flips: STAGE_BYTES is a multiple of 128, the XORs below touch bits 5-6)
    auto a_frag = [&](int i, int ty, int tx, int kk) -> f16x8 {
        return *(const f16x8*)(smem + (vt[tx] ^ (kk << 5)) + (i * 2 * HW_ + ty * HW_ + tx) * ROWB);
    };
    auto a_qfrag = [&](int i, int ty, int tx, int ms) -> v4i {
        const int base = SHIFT ? vq[ty * 2 + tx] : vq[tx];
        return *(const v4i*)(smem + (base ^ (ms << 5) ^ (SHIFT ? ((i & 1) << 5) : 0)) + (i * 2 * HW_ + ty * HW_ + tx) * QROWB);
    };
    auto flip_stage = [&](int delta) {
#pragma unroll
        for (int k = 0; k < 3; ++k) vt[k] += delta;
#pragma unroll
        for (int k = 0; k < (SHIFT ? 4 : 3); ++k) vq[k] += delta;
    };
    // E8M0 scales of the activation side: lanes 0-31 carry a_lo (stored * 2^c_lo_exp), lanes 32-63 a_hi (stored * 2^-c_hi_exp)
    const int a_scale = fh ? 127 + d.c_hi_exp : 127 - d.c_lo_exp;

    const int nchunks = d.Cin >> 6;
    // ---- prologue
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int u = 0; u < RING - 1; ++u) load_b(u, j, cur.tile_n, u);
        load_bq(j, cur.tile_n, 0);
    }
    if (NORM) {
        stage_norm(cur, 0);
        __syncthreads();
    }
    load_nf(0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) store_piece(0, r, load_piece(cur, 0, r));
    __syncthreads();
    Pend pend[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) pend[k] = load_piece(cur, 0, 0);      // (placeholder values: overwritten before their first use)

    f16x8 afr[AW];
    v4i aq[AW];
#pragma unroll
    for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0, 0);

    // The four waves run the same code and leave every barrier together, so their vector-memory instructions reach the CU's one address
    // unit in the same cycles and queue behind each other (measured: ~50 cycles of blocked issue per load, three of four waves waiting).
    // Wave w idles w * GDT_C_WSTAG * 64 cycles after each chunk barrier: the load clusters of the waves no longer coincide.
    auto wave_stagger = [&]() {
#if GDT_C_WSTAG > 0
        for (int k = 0; k < wave; ++k) __builtin_amdgcn_s_sleep(GDT_C_WSTAG);
#endif
    };
    int so = 0;                   // LDS offset of the halo stage of the current chunk (0 or STAGE_BYTES)
    int slot = 0;                 // (scale, shift) slot of the current tile
#ifdef GDT_C_STAMP                 // diagnostic build (tools/build_variant.sh stamp): per-wave s_memtime totals, written to d.stamp_out only
    unsigned long long st_body = 0, st_cbar = 0, st_tbar = 0, st_epi = 0, st_t = __builtin_amdgcn_s_memtime(), st_n = 0;
    const unsigned long long st_begin = st_t;
#define GDT_STAMP(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - st_t; st_t = now_; }
#else
#define GDT_STAMP(acc)
#endif
    for (;;) {
        const TileAt nxt = tile_at(vb + gridDim.x);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // One 64-channel chunk.  S2M (stride-2 form): the shifts with a non-zero weight block for the chunk's sub-pixel parity, a
        // COMPILE-TIME constant per instantiation of the body (dy = -1 needs odd input rows, dx = -1 odd columns: parity 0 -> shift 3
        // only, 1 -> 2 3, 2 -> 1 3, 3 -> all) -- the chunk loop below runs the four parities one after the other, so an inactive
        // tap costs neither MFMAs nor fragment / weight loads nor branches (with run-time masks: ~300 branches per chunk, and the
        // scheduler could not interleave the halo staging across them).
        auto chunk_body = [&](const int c, auto mask_tag) {
            constexpr unsigned S2M = decltype(mask_tag)::value;      // 0: run-time mask (256-column stride-2 tiles: four bodies spill 130-180 registers there)
            unsigned s2_rt = 0xFu;
            if (S2 && S2M == 0) { const int par = (c * 64) >> (d.lc8 + 1); s2_rt = (par & 2 ? 0xFu : 0xCu) & (par & 1 ? 0xFu : 0xAu); }
            auto s2_on = [&](int t) -> bool { return !S2 || (((S2M ? S2M : s2_rt) >> t) & 1u) != 0; };      // MFMAs of tap t
            auto s2_ld = [](int t) -> bool { return !S2 || S2M == 0 || ((S2M >> t) & 1u) != 0; };              // its loads (static masks only)
            const bool last = c + 1 == nchunks;
            const bool to_next = last && nxt.valid;
            const TileAt sta = to_next ? nxt : cur;
            const int sc = last ? 0 : c + 1, sslot = to_next ? slot ^ 1 : slot;
            if (NORM && nxt.valid && c == nchunks - 2) stage_norm(nxt, slot ^ 1);
            load_nf(sslot, sc);                 // (the table of the next tile was written during the previous chunk, a barrier ago)
            lo16 = lane_bytes(16); lo8 = lane_bytes(8); lo4 = lane_bytes(4);
            // (CT) which input shifts t have a non-zero weight block for column block j of this wave (gdt_ctf_column: 64-column
            // slices pair a cheap phase with an expensive one)
            // -- with gdt_ctc_column() column block j of EVERY wave is sub-pixel phase j of 32 output channels: the masks are
            // compile-time constants (phase 0: shift 0; 1: + dx; 2: + dy; 3: all four), the zero blocks cost neither MFMAs nor loads
            auto ct_on = [](int t, int j) -> bool { return !CT || !GDT_C_CT_SKIP || (((0xF531u >> (4 * j)) >> t) & 1u) != 0; };
            // tap of substep u (u >= SLOTS: the first substeps of the chunk staged now)
            auto t_of = [](int u) -> int { return u < SLOTS ? (u >> 2) : ((u - SLOTS) >> 2); };
            // (within the chunk: u < SLOTS; slices of the next chunk are always fetched -- its parity is another instantiation's business)
            auto s2_on_u = [&](int u) -> bool { return u >= SLOTS || s2_ld(u >> 2); };
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int ty = SHIFT ? (t >> 1) : t / 3, tx = SHIFT ? (t & 1) : t - ty * 3;
                const int nty = SHIFT ? ((t + 1) >> 1) : (t + 1) / 3, ntx = SHIFT ? ((t + 1) & 1) : (t + 1) - nty * 3;
                // k-substep index (fp16 array) and tile of substep u of this chunk, u >= SLOTS: the first substeps of the chunk staged now
                // (after the very last one this fetches the first slices again: unconditional loads keep the code straight-line)
                auto ks_of = [&](int u) -> long { return u < SLOTS ? (long)((u >> 2) * cin16 + c * 4 + (u & 3)) : (long)(sc * 4 + (u - SLOTS)); };
                auto tn_of = [&](int u) -> int { return (u >= SLOTS && last) ? nxt.tile_n : cur.tile_n; };
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int u = t * 4 + kk, cu = kk & 1;
                    if (!(GDT_C_ABL & 1) && u % SPR == 0) {      // halo of the next chunk: one loader round in flight, written SPR substeps after its load
                        const int r = u / SPR;
                        if (r >= DEPTH && r - DEPTH < NR) store_piece(STAGE_BYTES - so, r - DEPTH, pend[r % DEPTH]);
                        if (r < NR) pend[r % DEPTH] = load_piece(sta, sc, r);
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        // row block i: its MFMAs over the wave's columns share the activation fragment, which is then re-loaded IN PLACE
                        // for the next substep (single-buffered: 16 registers instead of 32)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            if (!(GDT_C_ABL & 8) && ct_on(t, j) && s2_on(t))
#ifdef GDT_C_SHAPE16_TIMING      // timing only (results are wrong): the same FLOPs, operand loads and accumulator registers on the 16 x 16 x 32 shape
                            {
                                typedef float f32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
                                for (int qq = 0; qq < 2; ++qq) {
                                    const int q4 = 4 * (2 * (u & 1) + qq);
                                    f32x4_ tq = {acc[i][j][q4], acc[i][j][q4 + 1], acc[i][j][q4 + 2], acc[i][j][q4 + 3]};
                                    tq = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[u % RING][j], afr[i % AW], tq, 0, 0, 0);
                                    acc[i][j][q4] = tq[0]; acc[i][j][q4 + 1] = tq[1]; acc[i][j][q4 + 2] = tq[2]; acc[i][j][q4 + 3] = tq[3];
                                }
                            }
#else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[u % RING][j], afr[i % AW], acc[i][j], 0, 0, 0);     // D[cout][pixel]
#endif
                        // this row's share of the substep's loads: column i of the ring slot substep u - 1 has finished with, (even
                        // substeps) the fp4 fragment and column i of the MX weights two groups ahead
                        if (!(GDT_C_ABL & 4) && i < TN && ct_on(t_of(u + RING - 1), i) && s2_on_u(u + RING - 1)) load_b((u + RING - 1) % RING, i, tn_of(u + RING - 1), ks_of(u + RING - 1));
                        // the window slot just used takes the fragment AW row blocks on: of this substep (AW < TM) or of the next one
                        if (GDT_C_ABL & 256) {}                                     // (timing only: no fp16 fragment re-loads)
                        else if (i + AW < TM) { if (s2_ld(t)) afr[i % AW] = a_frag(i + AW, ty, tx, kk); }
                        else if (kk < 3) { if (s2_ld(t)) afr[i % AW] = a_frag(i + AW - TM, ty, tx, kk + 1); }
                        else if (t < NTAP - 1) { if (s2_ld(t + 1)) afr[i % AW] = a_frag(i + AW - TM, nty, ntx, 0); }
                        if (cu == 0) {             // (the MX weights were last read at the end of substep u - 1; all columns are re-loaded
                            if (!(GDT_C_ABL & 64) && i < AW && s2_ld(t)) aq[i] = a_qfrag(i, ty, tx, kk >> 1);      //  behind the first rows: >= 24 MFMAs before their first use)
                            if (!(GDT_C_ABL & (2 | 128)) && 2 * i < TN && s2_ld(t)) {
                                if (ct_on(t_of(u), 2 * i)) load_bq(2 * i, tn_of(u), ks_of(u));
                                if (2 * i + 1 < TN && ct_on(t_of(u), 2 * i + 1)) load_bq(2 * i + 1, tn_of(u), ks_of(u));
                            }
                        }
                    }
                    if (!(GDT_C_ABL & 2) && cu == 1) {          // the correction product of the 32 k-values just done
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const v8i av = __builtin_shufflevector(aq[i % AW], aq[i % AW], 0, 1, 2, 3, -1, -1, -1, -1);
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                if (ct_on(t, j) && s2_on(t)) {
#ifdef GDT_C_SHAPE16_TIMING
                                    typedef float f32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
                                    for (int qq = 0; qq < 2; ++qq) {
                                        const int q4 = 4 * (2 * ((u >> 1) & 1) + qq);
                                        f32x4_ tq = {acc[i][j][q4], acc[i][j][q4 + 1], acc[i][j][q4 + 2], acc[i][j][q4 + 3]};
                                        tq = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(__builtin_shufflevector(bq[j], bq[j], 0, 1, 2, 3, 4, 5, -1, -1), av, tq, 2, 4, 0, bqs[j], 0, a_scale);
                                        acc[i][j][q4] = tq[0]; acc[i][j][q4 + 1] = tq[1]; acc[i][j][q4 + 2] = tq[2]; acc[i][j][q4 + 3] = tq[3];
                                    }
#else
                                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(__builtin_shufflevector(bq[j], bq[j], 0, 1, 2, 3, 4, 5, -1, -1), av, acc[i][j], 2, 4, 0, bqs[j], 0, a_scale);
#endif
                                }
                            if (!(GDT_C_ABL & 64) && i + AW < TM && s2_ld(t)) aq[i % AW] = a_qfrag(i + AW, ty, tx, kk >> 1);      // (AW < TM)
                        }
                    }
                    // in-order issue: lay the substep out as MFMA, a few VALU (the halo staging), MFMA, ... with the memory operations
                    // of a column behind its MFMAs
#if GDT_C_SCHED == 2
                    // one vector-memory instruction at a time: the L2 -> CU path takes ~1 KB per 30 cycles; a clump of loads backs
                    // the address unit up and the in-order wave cannot issue its next MFMA until the last of them is accepted
#pragma unroll
                    for (int m = 0; m < TM * TN; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                        if ((m & 1) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 global load per two MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 LDS read
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
                        if ((m & 3) == 3) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x040, 1, 0); }
                    }
                    if (cu == 1) {
#pragma unroll
                        for (int m = 0; m < TM * TN; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if ((m & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                            if (AW < TM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                            if ((m & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                        }
                    }
#elif !defined(GDT_C_NOSCHED)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
                        }
                        if (cu == 0 && j < 2) { __builtin_amdgcn_sched_group_barrier(0x020, 5, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }   // global loads, LDS reads
                        else if (cu == 0) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
                        else { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // LDS write
                        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);      // global store
                    }
                    if (cu == 1) {
#pragma unroll
                        for (int m = 0; m < TM * TN; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            GDT_STAMP(st_body)
            if (!last) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                wave_stagger();
                GDT_STAMP(st_cbar)
                flip_stage(STAGE_BYTES - 2 * so);
                so = STAGE_BYTES - so;
#pragma unroll
                for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0, 0);
            }
        };
        if (!S2) {
            for (int c = 0; c < nchunks; ++c) chunk_body(c, std::integral_constant<unsigned, 0xFu>());
        } else if (BN != 128) {
            for (int c = 0; c < nchunks; ++c) chunk_body(c, std::integral_constant<unsigned, 0u>());
        } else {
            const int cpp = nchunks >> 2;                   // chunks per sub-pixel parity (1 or 2)
            for (int cc = 0; cc < cpp; ++cc) chunk_body(cc, std::integral_constant<unsigned, 0x8u>());
            for (int cc = 0; cc < cpp; ++cc) chunk_body(cpp + cc, std::integral_constant<unsigned, 0xCu>());
            for (int cc = 0; cc < cpp; ++cc) chunk_body(2 * cpp + cc, std::integral_constant<unsigned, 0xAu>());
            for (int cc = 0; cc < cpp; ++cc) chunk_body(3 * cpp + cc, std::integral_constant<unsigned, 0xFu>());
        }

        // ------------------------------------------------------------ tile end: all waves are done with the last halo stage and
        // the next tile's first stage (written during the last chunk) is visible
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        GDT_STAMP(st_tbar)

        // ------------------------------------------------------------ epilogue, wave-private (no workgroup barrier).  MFMA operands
        // are swapped (D = W * A^T): lane (fr, fh) holds pixel fr of row block i and, in registers 4g .. 4g+3, the output channels
        // 8g + 4fh .. +3 of column block j.  Stored like that, one store instruction touches 32 different 128-byte lines with 32 bytes
        // each, and the address unit -- not HBM -- sets the pace (measured: 18 % of the kernel).  So every 32 x 32 block goes through a
        // 4 KB patch of the wave's own (XOR-swizzled, conflict-free both ways): written as it sits in the registers, read back with
        // 8 lanes per pixel, i.e. whole 128-byte lines per store instruction.  Statistics: a lane then owns 4 channels of 4 pixels per
        // block; the 8 pixel lanes are merged by a fixed butterfly.
        if (!(d.dbg & 4)) {
            float* __restrict__ outp = (float*)d.out;
            const float* __restrict__ resp = (const float*)d.res;
            float* patch = (float*)(smem + 2 * STAGE_BYTES + NORM_BYTES) + wave * (PIPE2 ? 2048 : 1024);
            int lane_e = lane_now();
            asm volatile("" : "+v"(lane_e));             // (opaque copy: keeps the epilogue's addresses out of the loop's invariant set)
            const int fr_e = lane_e & 31, fh_e = lane_e >> 5, pl = lane_e >> 3, q = lane_e & 7;
            const bool relu_now = d.relu != 0;
            const int wswz = ((fr_e >> 1) & 7) << 2;
            // Two bodies: tiles that lie wholly inside the output with no epilogue residual (every tile of the generator) run the
            // branch-free, software-pipelined body below; the checked body keeps the ragged edges and the residual read.
            const bool full_tile = (cur.y0 + 16 <= GH) & (cur.x0 + 16 <= GW) & (cur.tile_n * BN + wn * WTN + WTN <= d.Cout) & (resp == nullptr || (!CT && MODE == 0));
            auto epilogue_checked = [&]() {
                // pixel row (i, k) of this lane: y = y0 + 8 wm + 2 i + (k >> 1), x = x0 + 8 (k & 1) + pl -> one per-lane base offset plus a
                // uniform term per (i, k)
                const int yb = cur.y0 + wm * (WTM / 16), xb = cur.x0 + pl;
                const unsigned obase = CT ? (unsigned)((cur.n * d.OH + 2 * yb) * d.OW + 2 * xb) * (unsigned)d.phase_cout
                                          : (unsigned)((cur.n * GH + yb) * GW + xb) * (unsigned)d.Cout;
                auto roff = [&](int i, int k) -> unsigned {
                    return CT ? (unsigned)((2 * (2 * i + (k >> 1))) * d.OW + 16 * (k & 1)) * (unsigned)d.phase_cout
                              : (unsigned)((2 * i + (k >> 1)) * GW + 8 * (k & 1)) * (unsigned)d.Cout;
                };
                auto row_ok = [&](int i, int k) -> bool { return (yb + 2 * i + (k >> 1) < GH) & (xb + 8 * (k & 1) < GW); };
                constexpr int NH = TM / 4;              // 128-row statistics records per wave
                float st1[NH][4], st2[NH][4];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int colq = cur.tile_n * BN + wn * WTN + j * 32 + 4 * q;         // this lane's 4 channels after the transpose
                    const float4 bv = d.bias ? *(const float4*)(d.bias + colq) : make_float4(0.f, 0.f, 0.f, 0.f);
                    int ct_ph = 0, ct_co = 0;
                    if (CT) gdt_ctc_column(colq, ct_ph, ct_co);
                    const unsigned coff = CT ? (unsigned)(((ct_ph >> 1) * d.OW + (ct_ph & 1)) * d.phase_cout + ct_co) : (unsigned)colq;
                    if (!CT || j == 0) {
#pragma unroll
                        for (int h = 0; h < NH; ++h)
#pragma unroll
                            for (int e = 0; e < 4; ++e) { st1[h][e] = 0.f; st2[h][e] = 0.f; }
                    }
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x16& a = acc[i][j];
                            *(float4*)(patch + fr_e * 32 + ((8 * g + 4 * fh_e) ^ wswz)) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int row = 8 * k + pl;
                            float4 v = *(const float4*)(patch + row * 32 + ((4 * q) ^ (((row >> 1) & 7) << 2)));
                            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                            const unsigned o = obase + roff(i, k) + coff;
                            if (row_ok(i, k) & (colq < d.Cout)) {
                                if (d.stats) {
                                    st1[i / 4][0] += v.x; st1[i / 4][1] += v.y; st1[i / 4][2] += v.z; st1[i / 4][3] += v.w;
                                    st2[i / 4][0] += v.x * v.x; st2[i / 4][1] += v.y * v.y; st2[i / 4][2] += v.z * v.z; st2[i / 4][3] += v.w * v.w;
                                }
                                if (!CT && resp) { const float4 rv = *(const float4*)(resp + o); v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w; }
                                if (relu_now) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                                *(float4*)(outp + o) = v;
                            }
                        }
                    }
                    if (d.stats && (!CT || j == TN - 1)) {        // (CT: the wave's four blocks are the four phases of the same channels)
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
#pragma unroll
                                for (int msk = 8; msk < 64; msk <<= 1) { st1[h][e] += __shfl_xor(st1[h][e], msk); st2[h][e] += __shfl_xor(st2[h][e], msk); }
                            if (pl == 0 && colq < d.Cout) {
                                const int rec = cur.tile_m * (BM / 128) + wm * NH + h;          // 128-row record of the patch
                                float* dst;
                                if (!CT) dst = d.stats + ((long)(d.stats_tile_base + rec) * 2) * d.Cout + colq;
                                else dst = d.stats + ((long)rec * 2) * d.phase_cout + ct_co;
                                const int cstride = CT ? d.phase_cout : d.Cout;
#pragma unroll
                                for (int e = 0; e < 4; ++e) { dst[e] = st1[h][e]; dst[cstride + e] = st2[h][e]; }
                            }
                        }
                    }
                }
            };
            // The interior body of the 3x3 and stride-2 forms, software-pipelined: two patches per wave, block n + 1 is written while block n's
            // four row reads are in flight (one LDS latency per block instead of four), the store offset is a running per-lane value
            // (o, o + 8 Cout, o + W Cout, o + W Cout + 8 Cout, then o += 2 W Cout: three uniform constants instead of one hoisted --
            // and spilled -- scalar per pixel row), the statistics of the wave's 256 pixels go into ONE record (the second one is
            // written as zeros) and the 8 pixel lanes are merged by a DPP rotate, a swizzle and one permute per value.
            auto epilogue_pipelined = [&]() {
                const int yb = cur.y0 + wm * (WTM / 16), xb = cur.x0 + pl;
                // one output row pair down / 8 pixels to the right, in elements (transposed form: input pixel (y, x) owns outputs (2y + py, 2x + px))
                const unsigned rowstep = CT ? 2u * (unsigned)d.OW * (unsigned)d.phase_cout : (unsigned)GW * (unsigned)d.Cout;
                const unsigned c8 = CT ? 16u * (unsigned)d.phase_cout : 8u * (unsigned)d.Cout;
                const float lo = relu_now ? 0.f : -__builtin_inff();
                auto put = [&](int blk) {
                    const f32x16& a = acc[blk % TM][blk / TM];
                    float* pw = patch + (PIPE2 ? (blk & 1) * 1024 : 0);      // (one patch: the wave's LDS operations execute in order, so the next block's writes land behind this block's reads)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *(float4*)(pw + fr_e * 32 + ((8 * g + 4 * fh_e) ^ wswz)) = make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                };
                put(0);
                // epilogue residual (BatchNorm generators: y = x + conv_block(x), p2p_networks.py:505 with the norm folded into the conv): the four lines of a block
                // are requested one block ahead of their use, like the patch writes (in the checked body they were fetched and awaited inside the store loop:
                // 0.56 vs 0.41 ms per conv)
                constexpr bool EPI_RES = !CT && MODE == 0;
                float4 rcur[4], rnxt[4];
                auto fetch_res = [&](float4 (&r)[4], unsigned oo) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k] = *(const float4*)(resp + (oo + ((k & 1) ? c8 : 0u) + ((k >> 1) ? rowstep : 0u)));
                };
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int colq = cur.tile_n * BN + wn * WTN + j * 32 + 4 * q;
                    const float4 bv = d.bias ? *(const float4*)(d.bias + colq) : make_float4(0.f, 0.f, 0.f, 0.f);
                    int ct_ph = 0, ct_co = 0;
                    if (CT) gdt_ctc_column(colq, ct_ph, ct_co);
                    unsigned o = CT ? (unsigned)((cur.n * d.OH + 2 * yb + (ct_ph >> 1)) * d.OW + 2 * xb + (ct_ph & 1)) * (unsigned)d.phase_cout + (unsigned)ct_co
                                    : (unsigned)((cur.n * GH + yb) * GW + xb) * (unsigned)d.Cout + (unsigned)colq;
                    if (!CT) {               // (CT: the wave's four blocks are the four phases of the same channels: one record over all of them)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
                    }
                    if (EPI_RES && resp) fetch_res(rcur, o);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int blk = j * TM + i;
                        const float* pr = patch + (PIPE2 ? (blk & 1) * 1024 : 0);
                        float4 v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int row = 8 * k + pl;
                            v[k] = *(const float4*)(pr + row * 32 + ((4 * q) ^ (((row >> 1) & 7) << 2)));
                        }
                        if (blk + 1 < TM * TN) put(blk + 1);
                        asm volatile("" : "+v"(o));
                        if (EPI_RES && resp && i + 1 < TM) fetch_res(rnxt, o + 2u * rowstep);      // next block of this column (the column's first: above)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float4 t = v[k];
                            t.x += bv.x; t.y += bv.y; t.z += bv.z; t.w += bv.w;
                            s1[0] += t.x; s1[1] += t.y; s1[2] += t.z; s1[3] += t.w;
                            s2[0] += t.x * t.x; s2[1] += t.y * t.y; s2[2] += t.z * t.z; s2[3] += t.w * t.w;
                            if (EPI_RES && resp) { t.x += rcur[k].x; t.y += rcur[k].y; t.z += rcur[k].z; t.w += rcur[k].w; }
                            t.x = fmaxf(t.x, lo); t.y = fmaxf(t.y, lo); t.z = fmaxf(t.z, lo); t.w = fmaxf(t.w, lo);
                            *(float4*)(outp + (o + ((k & 1) ? c8 : 0u) + ((k >> 1) ? rowstep : 0u))) = t;
                        }
                        o += 2u * rowstep;
                        if (EPI_RES && resp) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) rcur[k] = rnxt[k];
                        }
                    }
                    if (d.stats && (!CT || j == TN - 1)) {
                        auto merge = [](float x) -> float {            // sum over the 8 lanes lane ^ {8, 16, 32}, fixed order
                            x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));    // row_ror:8
                            x += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), 0x401F));                        // lane ^ 16
                            x += __shfl_xor(x, 32);
                            return x;
                        };
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1[e] = merge(s1[e]); s2[e] = merge(s2[e]); }
                        if (pl == 0) {
                            constexpr int NH = TM / 4;
                            const int rec = cur.tile_m * (BM / 128) + wm * NH;
                            const int cstride = CT ? d.phase_cout : d.Cout;
                            float* dst = CT ? d.stats + ((long)rec * 2) * d.phase_cout + ct_co : d.stats + ((long)(d.stats_tile_base + rec) * 2) * d.Cout + colq;
                            *(float4*)dst = make_float4(s1[0], s1[1], s1[2], s1[3]);
                            *(float4*)(dst + cstride) = make_float4(s2[0], s2[1], s2[2], s2[3]);
#pragma unroll
                            for (int h = 1; h < NH; ++h) {
                                *(float4*)(dst + (long)h * 2 * cstride) = make_float4(0.f, 0.f, 0.f, 0.f);
                                *(float4*)(dst + (long)h * 2 * cstride + cstride) = make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                        }
                    }
                }
            };
            if (full_tile) epilogue_pipelined();
            else epilogue_checked();
        } else {
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) sacc += acc[i][j][0] + acc[i][j][15];
            if (sacc == 12345.678f) ((float*)d.out)[0] = sacc;
        }
#ifdef GDT_C_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (diagnostic only: the epilogue's stores are charged to the epilogue)
        GDT_STAMP(st_epi)
        ++st_n;
#endif
        if (!nxt.valid) break;
        cur = nxt; vb += gridDim.x; slot ^= 1;
        flip_stage(STAGE_BYTES - 2 * so);
        so = STAGE_BYTES - so;
#pragma unroll
        for (int i = 0; i < AW; ++i) afr[i] = a_frag(i, 0, 0, 0);
    }
#ifdef GDT_C_STAMP
    if (lane == 0 && d.stamp_out) {
        unsigned long long* o = d.stamp_out + ((long)blockIdx.x * NWAVES + wave) * 8;
        o[0] = st_body; o[1] = st_cbar; o[2] = st_tbar; o[3] = st_epi; o[4] = __builtin_amdgcn_s_memtime() - st_begin; o[5] = st_n;
    }
#endif
}

// timing-only ablation knob: GDT_C_DBG=4 skips the epilogue
static int c_dbg() { static const int v = [] { const char* e = getenv("GDT_C_DBG"); return e ? atoi(e) : 0; }(); return v; }

// Four waves, one per SIMD (512 registers each: 256 accumulator AGPRs + 256 VGPRs).  TALL = false: 2 x 2 waves of 128 pixels x 128
// channels; TALL = true: 1 x 4 waves of 256 pixels x 64 channels -- every wave reads all 256 pixels' fragments from LDS (twice the
// LDS bytes per MFMA) and streams a quarter of the weights from L1 (half the L1 bytes per MFMA: the L1 weight stream is the scarcer
// of the two, ~60 % of its bandwidth in the square shape).  Measured on the resblock conv: 0.422 -> 0.408 ms plain, 0.518 -> 0.485
// with the residual + write-back staging.  (An eight-wave form, two waves of 128 x 64 per SIMD at 256 registers, runs its fp16
// core faster -- 0.185 vs 0.243 ms -- but pays 0.09 ms for the weight stream and 0.07 for the staging: 0.443 ms complete.)
// BN_ = 128 (output channel counts that are not a multiple of 256: the 64 -> 128 stride-2 layer): 2 x 2 waves of 128 pixels x 64 channels.
// W8 (with TALL): eight waves, two per SIMD at 256 registers each, 1 x 8: every wave owns all 256 pixels x 32 channels.  The same
// weight bytes per MFMA as 1 x 4 (a weight fragment still feeds 8 MFMAs), twice its LDS fragment bytes (an activation fragment feeds one
// MFMA); what it buys is a second wave per SIMD: while one waits -- for a weight fragment behind older halo loads (vmcnt retires in
// order), for an LDS fragment, at the address unit -- the other issues.
// W8 with BN_ = 128 (the 64 -> 128 stride-2 layer): eight waves as 2 x 4, a wave = 128 pixels x 32 channels.  That layer has little matrix work per halo byte and its
// four-wave form spends a tile adding up what ONE wave per SIMD must issue in order: staging arithmetic, loads, fragment reads, MFMAs.
template <int MODE, int FORM = 0, bool TALL = false, int BN_ = 256, bool W8 = false>
int launch_c(const ConvLaunch& d, hipStream_t stream) {
    constexpr int BN = BN_, WGM = TALL ? 1 : 2, WGN = TALL ? (W8 ? 8 : 4) : (W8 ? 4 : 2);
    constexpr size_t LDS_BYTES = lds_bytes(WGM * WGN);
    static_assert(BN == 256 || (BN == 128 && !TALL), "tile width");
    // (the transposed form's compile-time skipping of zero weight blocks assumes that a wave's column blocks ARE the four sub-pixel phases, i.e. four blocks per wave:
    //  with 2 x 4 waves the phases of a wave depend on its column -- not instantiated; the 256-column stride-2 form spills at 256 registers: 0.35 -> 0.61 ms)
    static_assert(!W8 || (TALL && FORM == 0) || (!TALL && MODE == 1 && FORM == 2 && BN == 128), "the eight-wave layouts: 1 x 8 for the 3x3 form; 2 x 4 for the 128-column stride-2 form");
    const int gh = FORM == 2 ? d.OH : d.H, gw = FORM == 2 ? d.OW : d.W;
    const int tiles = d.N * ((gw + 15) / 16) * ((gh + PH - 1) / PH), ntn = d.CoutPad / BN;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_c_kernel<BN, WGM, WGN, MODE, FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
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
    constexpr int NWV = WGM * WGN;
    if (!stamp_buf) GDT_CHECK_HIP(hipMalloc((void**)&stamp_buf, (size_t)cus * NWV * 8 * sizeof(unsigned long long)));
    GDT_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, (size_t)cus * NWV * 8 * sizeof(unsigned long long), stream));
    ds.stamp_out = stamp_buf;
    hipLaunchKernelGGL((conv3x3_halo_c_kernel<BN, WGM, WGN, MODE, FORM>), dim3(grid), dim3(WGM * WGN * 64), LDS_BYTES, stream, ds, vblocks);
    if (++stamp_calls % 200 < 20) {          // a few launches per variant of a sustained run
        GDT_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)grid * NWV * 8);
        GDT_CHECK_HIP(hipMemcpy(h.data(), stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s[6] = {0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)grid * NWV; ++w) for (int k = 0; k < 6; ++k) s[k] += (double)h[w * 8 + k];
        const double nw = grid * (double)NWV, nt = s[5] / nw;
        fprintf(stderr, "[c stamp] MODE %d FORM %d BN %d waves %d: tiles/wave %.1f; per tile: chunk bodies %.0f, chunk barriers %.0f, tile barrier %.0f, epilogue %.0f cycles; total per wave %.0f\n",
                MODE, FORM, BN, NWV, nt, s[0] / nw / nt, s[1] / nw / nt, s[2] / nw / nt, s[3] / nw / nt, s[4] / nw);
    }
#else
    hipLaunchKernelGGL((conv3x3_halo_c_kernel<BN, WGM, WGN, MODE, FORM>), dim3(grid), dim3(WGM * WGN * 64), LDS_BYTES, stream, d, vblocks);
#endif
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// Eligible: 3x3 / stride 1 / pad 1 (zero or reflect), Cin % 64 == 0, 256-wide output tiles, the fp16 + MX fragment-ordered weights
// present, whole patches when statistics are taken, enough patches to fill the chip; with a folded InstanceNorm 128 <= Cin <= 256.
bool gdt_conv_halo_c_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_HALO_C"); return e ? atoi(e) : 1; }();   // 0 off, 1 auto, 2 force
    if (mode == 0 || !d.w_cfrag || !d.wmx_a || !d.wmx_b || !d.wmx_s) return false;
    const bool shape = d.ntaps == 9 && d.TW == 3 && d.sy == 1 && d.sx == 1 && d.dy0 == -1 && d.dx0 == -1 && d.dys == 1 && d.dxs == 1 &&
                       d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0 && d.Cin % 64 == 0 && !d.out_f32 && d.Cout % 8 == 0 &&
                       d.OH == d.H && d.OW == d.W && d.Kpad == 9 * d.Cin && d.CoutPad % 256 == 0 && !d.pool2 && !d.phase_cout;
    if (!shape) return false;
    if (d.in_norm && (d.Cin > 256 || d.Cin < 128)) return false;
    if ((d.in_res || d.in_out) && !d.in_norm) return false;
    if (d.stats && ((d.H & 15) || (d.W & 15))) return false;
    if ((long)d.N * d.H * d.W * d.Cin >= (1L << 30) || (long)d.N * d.H * d.W * d.Cout >= (1L << 32)) return false;
    if (mode == 2) return true;
    const long tiles = (long)d.N * ((d.W + 15) / 16) * ((d.H + 15) / 16);
    const double useful = (double)d.H * d.W / ((double)((d.H + 15) / 16 * 16) * ((d.W + 15) / 16 * 16));
    static const int min_tiles = [] { const char* e = getenv("GDT_CONV_MIN_TILES"); return e ? atoi(e) : 16; }();      // (below: the generic f16x3 kernels; batch 1-4 at 256^2 measured 2.09 vs 2.38 ms with the patch kernels on 16 tiles)
    return tiles * (d.CoutPad / 256) >= min_tiles && useful >= 0.85;
}

// few patches (batch 1-4 at 256^2: 16-64 tiles for 256 CUs): 128-column tiles double the number of workgroups
int gdt_conv_halo_c_columns(const ConvLaunch& d) {
    static const int narrow_below = [] { const char* e = getenv("GDT_C_NARROW_BELOW"); return e ? atoi(e) : 192; }();
    const long tiles256 = (long)d.N * ((d.W + 15) / 16) * ((d.H + PH - 1) / PH) * (d.CoutPad / 256);
    return tiles256 < narrow_below ? 128 : 256;
}

int gdt_launch_conv_halo_c(const ConvLaunch& d_in, hipStream_t stream) {
    static const int dbg = [] { const char* e = getenv("GDT_RB_DBG"); return e ? atoi(e) : 0; }();
    ConvLaunch d = d_in;
    d.dbg = dbg;
    static const int stagger = [] { const char* e = getenv("GDT_C_STAGGER_US"); return e ? atoi(e) : 0; }();
    d.stagger_us = stagger;
#ifdef GDT_C_DEV_W8_ONLY      // dev builds (resource reports, asm): only the eight-wave instantiations are compiled
    {
        if (!d.in_norm) return launch_c<0, 0, true, 256, true>(d, stream);
        if (d.in_res) return d.in_out ? launch_c<7, 0, true, 256, true>(d, stream) : launch_c<3, 0, true, 256, true>(d, stream);
        return d.in_out ? launch_c<5, 0, true, 256, true>(d, stream) : launch_c<1, 0, true, 256, true>(d, stream);
    }
#else
    if (gdt_conv_halo_c_columns(d) == 128) {
        if (!d.in_norm) return launch_c<0, 0, false, 128>(d, stream);
        if (d.in_res) {
            if (d.in_out) return launch_c<7, 0, false, 128>(d, stream);
            return launch_c<3, 0, false, 128>(d, stream);
        }
        return d.in_out ? launch_c<5, 0, false, 128>(d, stream) : launch_c<1, 0, false, 128>(d, stream);
    }
    static const int tall = [] { const char* e = getenv("GDT_C_TALL"); return e ? atoi(e) : 1; }();      // 0: the 2 x 2 wave layout
    static const int waves8 = [] { const char* e = getenv("GDT_C_WAVES"); return e ? atoi(e) == 8 : false; }();      // 8: the 1 x 8 layout, two waves per SIMD
    if (tall && waves8) {
        if (!d.in_norm) return launch_c<0, 0, true, 256, true>(d, stream);
        if (d.in_res) {
            if (d.in_out) return launch_c<7, 0, true, 256, true>(d, stream);
            return launch_c<3, 0, true, 256, true>(d, stream);
        }
        return d.in_out ? launch_c<5, 0, true, 256, true>(d, stream) : launch_c<1, 0, true, 256, true>(d, stream);
    }
    if (tall) {
        if (!d.in_norm) return launch_c<0, 0, true>(d, stream);
        if (d.in_res) {
            if (d.in_out) return launch_c<7, 0, true>(d, stream);
            return launch_c<3, 0, true>(d, stream);
        }
        return d.in_out ? launch_c<5, 0, true>(d, stream) : launch_c<1, 0, true>(d, stream);
    }
    if (!d.in_norm) return launch_c<0>(d, stream);
    if (d.in_res) {
        if (d.in_out) return launch_c<7>(d, stream);
        return launch_c<3>(d, stream);
    }
    return d.in_out ? launch_c<5>(d, stream) : launch_c<1>(d, stream);
#endif
}

// Transposed form (CT): ConvTranspose2d(k3,s2,p1,op1) as one launch (phase_cout > 0, weights of Op::ctf packed by pack_mx), 64 or 128
// channels per phase, enough patches to fill the chip, at most 15 % padding waste; with statistics whole 16x16 patches; a folded
// InstanceNorm needs 128 <= Cin <= 256 (table slots, staged one chunk ahead).
bool gdt_conv_halo_c_ct_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_HALO_C"); return e ? atoi(e) : 1; }();   // 0 off
    if (mode == 0 || !d.phase_cout || !d.w_cfrag || !d.wmx_a || !d.wmx_b || !d.out || d.out_f32 || d.res || d.in_out) return false;
    if ((d.phase_cout != 64 && d.phase_cout != 128) || d.Cout != 4 * d.phase_cout || d.CoutPad != d.Cout || d.Cin % 64 != 0) return false;
    if (d.ntaps != 4 || d.TW != 2 || d.Kpad != 4 * d.Cin || d.pad_reflect) return false;
    if (d.in_norm && (d.Cin > 256 || d.Cin < 128)) return false;
    if (d.in_res && !d.in_norm) return false;
    if (d.stats && ((d.H & 15) || (d.W & 15))) return false;
    if ((long)d.N * d.H * d.W * d.Cin >= (1L << 30) || (long)d.N * d.OH * d.OW * d.phase_cout >= (1L << 32)) return false;
    const long tiles = (long)d.N * ((d.W + 15) / 16) * ((d.H + 15) / 16);
    const double useful = (double)d.H * d.W / ((double)((d.H + 15) / 16 * 16) * ((d.W + 15) / 16 * 16));
    static const int min_tiles = [] { const char* e = getenv("GDT_CONV_MIN_TILES"); return e ? atoi(e) : 16; }();      // (below: the generic f16x3 kernels; batch 1-4 at 256^2 measured 2.09 vs 2.38 ms with the patch kernels on 16 tiles)
    return tiles * (d.CoutPad / 256) >= min_tiles && useful >= 0.85;
}

int gdt_launch_conv_halo_c_ct(const ConvLaunch& d_in, hipStream_t stream) {
#ifdef GDT_C_DEV_W8_ONLY
    return GDT_ERR_INVALID;
#else
    ConvLaunch d = d_in;
    d.dbg = c_dbg();
    if (!d.in_norm) return launch_c<0, 1>(d, stream);
    return d.in_res ? launch_c<3, 1>(d, stream) : launch_c<1, 1>(d, stream);
#endif
}

// Stride-2 form (S2): Conv2d(k3, s2, p1, zero padding) over the virtual space-to-depth view of its input.  The launch descriptor carries
// the VIRTUAL channel count (Cin = 4 x real, lc8 likewise, Kpad = 16 x real, 4 "taps" = input shifts; weights of Op::s2), H / W the real
// input size, OH / OW the output the patches tile.  Real Cin 64 or 128 (a 64-channel chunk must not straddle two sub-pixel parities; a folded
// InstanceNorm keeps its table of the REAL channels in the 256-entry slots), 256-wide output tiles (Cout 128 is padded with zero columns).
bool gdt_conv_halo_c_s2_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_HALO_C"); return e ? atoi(e) : 1; }();   // 0 off
    if (mode == 0 || !d.w_cfrag || !d.wmx_a || !d.wmx_b || !d.out || d.out_f32 || d.res || d.phase_cout || d.pool2 || d.pad_reflect) return false;
    const int cr = d.Cin >> 2;
    if ((cr != 64 && cr != 128) || d.ntaps != 4 || d.TW != 2 || d.Kpad != 4 * d.Cin || d.CoutPad % 128 != 0 || d.Cout % 8 != 0) return false;
    if (d.OH != (d.H - 1) / 2 + 1 || d.OW != (d.W - 1) / 2 + 1) return false;
    if (d.in_res || (d.in_out && !d.in_norm)) return false;      // no residual mode in the stride-2 launch: a residual would be dropped silently
    if (d.stats && ((d.OH & 15) || (d.OW & 15))) return false;
    if ((long)d.N * d.H * d.W * cr >= (1L << 30) || (long)d.N * d.OH * d.OW * d.Cout >= (1L << 32)) return false;
    const long tiles = (long)d.N * ((d.OW + 15) / 16) * ((d.OH + 15) / 16);
    const double useful = (double)d.OH * d.OW / ((double)((d.OH + 15) / 16 * 16) * ((d.OW + 15) / 16 * 16));
    static const int min_tiles = [] { const char* e = getenv("GDT_CONV_MIN_TILES"); return e ? atoi(e) : 16; }();      // (below: the generic f16x3 kernels; batch 1-4 at 256^2 measured 2.09 vs 2.38 ms with the patch kernels on 16 tiles)
    return tiles * (d.CoutPad % 256 == 0 ? d.CoutPad / 256 : d.CoutPad / 128) >= min_tiles && useful >= 0.85;
}

int gdt_launch_conv_halo_c_s2(const ConvLaunch& d_in, hipStream_t stream) {
#ifdef GDT_C_DEV_W8_ONLY
    return GDT_ERR_INVALID;
#else
    ConvLaunch d = d_in;
    d.dbg = c_dbg();
    if (d.CoutPad % 256 != 0) {
        static const int w8s2 = [] { const char* e = getenv("GDT_C_S2_WAVES"); return e ? atoi(e) == 8 : true; }();      // two waves per SIMD (2 x 4): 0.55 -> 0.51 ms; GDT_C_S2_WAVES=4: back to four
        if (w8s2 && d.in_norm && !d.in_out) return launch_c<1, 2, false, 128, true>(d, stream);
        if (!d.in_norm) return launch_c<0, 2, false, 128>(d, stream);
        return d.in_out ? launch_c<5, 2, false, 128>(d, stream) : launch_c<1, 2, false, 128>(d, stream);
    }
    static const int tall = [] { const char* e = getenv("GDT_C_TALL_S2"); return e ? atoi(e) : 1; }();      // 0: the 2 x 2 wave layout
    if (tall) {
        if (!d.in_norm) return launch_c<0, 2, true>(d, stream);
        return d.in_out ? launch_c<5, 2, true>(d, stream) : launch_c<1, 2, true>(d, stream);
    }
    if (!d.in_norm) return launch_c<0, 2>(d, stream);
    return d.in_out ? launch_c<5, 2>(d, stream) : launch_c<1, 2>(d, stream);
#endif
}
