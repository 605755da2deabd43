// Generator head: 7x7 / stride-1 / pad-3 convolution from 64 channels to <= 4 image channels with tanh, fp32 NCHW output
// (p2p_networks.py:433-436: ReflectionPad2d(3), Conv2d(ngf, output_nc, 7), Tanh) -- one fused, persistent kernel.
//
// A 64 -> 3 channel conv wastes the 32-wide MFMA N dimension; the row-split form fixes that: the GEMM runs over the 7 kernel
// ROWS only (K = 7 * 64) and produces, per pixel, the 7 * cout partial sums P[kx * cout + co] (N = 21 -> 32), and the output
// is the sum of 7 horizontally shifted partials.  The first implementation did this with the generic implicit GEMM plus a
// combine kernel: every input row went L2 -> LDS seven times (once per kernel row) and the [pixels][24] fp16 partials made a
// round trip through HBM -- 0.87 ms for 79 GFLOP, the slowest layer of the generator.  Here:
//   * one workgroup computes an 8 x 32 output tile: the 14 x 38 input halo is staged ONCE in LDS (LDS-DMA, reflect padding
//     resolved in the source address) and, when the preceding InstanceNorm + ReLU is folded in, normalised in place;
//   * 4 wavefronts share the ten 32-row blocks of the 8 x 38 partial-sum image (the partials are needed 3 columns past the
//     tile on either side): 28 v_mfma_f32_32x32x16_f16 per block with the whole weight matrix resident in registers (28 B
//     fragments, loaded once: the kernel is persistent);
//   * the fp32 partials go to LDS (over the consumed halo), and the combine (7 taps, + bias, tanh) writes fp32 NCHW rows of
//     32 pixels.  Nothing but the input and the image touches HBM (537 MB + 50 MB per 64 x 256^2 batch);
//   * 70 KB of LDS and 4 waves: two workgroups per CU, so one loads / normalises / combines while the other runs its MFMAs
//     (a single double-buffered workgroup measured 0.46 ms with its phases serialised by barriers).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gdt_common.h"

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int PH = 8, PW = 32, KT = 7;
constexpr int HWD = PW + KT - 1;                 // 38 halo columns
constexpr int HHT = PH + KT - 1;                 // 14 halo rows
constexpr int HPIX = HHT * HWD;                  // 532 halo pixels of 128 B
constexpr int NWAVE = 4, NT = NWAVE * 64, NBLK = 10;        // ten 32-row blocks of partial-sum pixels over four waves
constexpr int MROWS = PH * HWD;                  // 304 partial-sum pixels
constexpr int LOAD_ROWS8 = (HPIX + 7) / 8;       // 67 wave-wide 1 KB loads per tile
constexpr int LROWS = 552;                       // LDS rows of the halo buffer: >= NBLK * 32 - 1 + 6 * HWD + 1 = 548
constexpr int HBYTES = LROWS * 128;              // 70,656 B: two workgroups per CU
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROUNDS = (HPIX + NT / 8 - 1) / (NT / 8);      // 17 staging rounds of 32 halo pixels (F32IN form)
constexpr int PSTRIDE = 33;                      // floats per partial-sum pixel (odd: the 7-tap combine is conflict free)
static_assert(NBLK * 32 >= MROWS && NBLK * 32 * PSTRIDE * 4 <= HBYTES && LROWS >= LOAD_ROWS8 * 8 && NBLK <= 3 * NWAVE && ROUNDS * (NT / 8) <= LROWS, "layout");

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_dst, 16, 0, 0);
}
// LDS-only workgroup barrier (no global-memory fence: global stores / loads stay in flight across it)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// F32IN ("f16c" precision mode): the input is the fp32 NHWC tensor of that mode; the halo goes through registers (load, folded
// InstanceNorm + ReLU in fp32, split into fp16(a) and the residual) instead of LDS-DMA + in-place normalisation.  With the
// block-scaled correction operands present (d.wmx_a: packed by net.hip for the f16c head) the product is compensated like the other
// layers of the mode (conv3x3_halo_c.hip):  a w ~= a_hi w_hi (fp16 MFMA) + [a_lo | a_hi]_fp4 . [w_hi | w_lo]_fp6 (one MX MFMA per 32
// k-values).  LDS holds one plane at a time -- two workgroups per CU must keep fitting -- so the fp4 words (two dwords per piece, 34
// registers per thread) wait in registers while the fp16 plane is consumed, are then written over it, and a second, shorter MFMA
// pass (14 instead of 28 per block, weights streamed from L2) adds the correction.  A single fp16 pass left the head with 3.4e-4 of
// the output range -- as much as the 22 compensated layers before it together (4.7e-4 pre-tanh overall; 3.1e-4 with this).
//
// X3 ("f16x3" precision mode, round 5): the exact split  a w = a_hi w_hi + (a_lo w_hi + a_hi w_lo) 2^-11  (conv_igemm_x3.hip) on the same skeleton.  LDS still
// holds one plane at a time, and the lo plane (fp16: 68 registers per thread) cannot wait in registers beside two accumulator sets: the halo is read TWICE -- the
// second time, microseconds later, from L2 / the Infinity Cache -- and split again; pass 1 runs a_hi against w_hi and w_lo (both streamed L2 -> registers), pass 2
// a_lo against w_hi.  Replaces the generic K-step-32 GEMM + partial-sum tensor + combine launch of the mode (0.96 + 0.67 ms at 64 x 256^2) and lets the planner
// fold the InstanceNorm in front of the head (one more 0.4 ms pass over the 1 GB tensor).
template <bool F32IN, bool MX = false, bool X3 = false>
__global__ __launch_bounds__(NT, 2) void conv_head7_kernel(const ConvLaunch d, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int tiles_x = (d.W + PW - 1) / PW, tiles_y = (d.H + PH - 1) / PH, tpi = tiles_x * tiles_y;
    const int cout = d.Cout;
    static_assert((F32IN || !MX) && (!X3 || (F32IN && !MX)), "the compensated / exact products read the fp32 tensor");
    constexpr bool STREAM = MX || X3;           // weight fragments streamed per tile instead of resident

    // XCD-chunked persistent schedule: workgroup b (XCD b & 7, slot b >> 3) walks the tiles slot, slot + S, ... of its XCD's
    // contiguous span, so the CUs of one XCD hold neighbouring tiles (shared halo rows hit its L2) at any time
    const int per_xcd = (ntiles + 7) >> 3, S = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int span_lo = xcd * per_xcd, span_hi = min(span_lo + per_xcd, ntiles);
    int tile = span_lo + slot;
    if (tile >= span_hi) return;

    // Blocks of 32 partial-sum pixels per wave: 10 blocks over 4 waves is 3,3,2,2.  The two workgroups that share a CU sit
    // on the same four SIMDs; rotating the assignment by two waves in every other workgroup makes it 5,5,5,5 per SIMD.
    const int wrot = (wave + 2 * (int)(blockIdx.x >= gridDim.x / 2)) & 3;     // (workgroups b and b + grid/2 share a CU)
    const int nb = wrot < NBLK - 2 * NWAVE ? 3 : 2;         // blocks wrot, wrot + 4 (, wrot + 8)

    // the whole B operand lives in registers: fragment ks holds k = ks * 16 + fh * 8 .. +8 of output column fr
    // (MX: the registers are needed for the fp4 words that wait for the second pass; the 28 KB matrix is then streamed L2 -> registers
    // WRING fragments ahead, once per tile -- every wave of the chip reads the same 28 KB)
    constexpr int NBW = STREAM ? 1 : 4 * KT, WRING = X3 ? 4 : 6;
    f16x8 bw[NBW];
    if (!STREAM) {
#pragma unroll
        for (int ks = 0; ks < 4 * KT; ++ks) bw[ks % NBW] = *(const f16x8*)(d.w_frag + ((long)ks * 64 + lane) * 8);
    }
    f16x8 wring[STREAM ? WRING : 1], wring_lo[X3 ? WRING : 1];
    // (uniform base + 32-bit lane offset, refreshed per tile behind an opaque copy: as 64-bit per-lane addresses the 28 + 42 fragment
    // addresses are loop invariants, which the compiler hoists out of the tile loop and spills)
    unsigned lo16 = 0, lo8 = 0, lo4 = 0;
    auto lane_offsets = [&]() { int l = lane; asm volatile("" : "+v"(l)); lo16 = l * 16; lo8 = l * 8; lo4 = l * 4; };
    auto load_w = [&](int ks) { wring[ks % WRING] = *(const f16x8*)((const char*)d.w_frag + ks * 1024 + lo16); };
    auto load_wl = [&](int ks) { wring_lo[X3 ? ks % WRING : 0] = *(const f16x8*)((const char*)d.w_frag2 + ks * 1024 + lo16); };       // (X3: d.w_frag2 = the lo parts, same order)

    const bool refl = d.pad_reflect != 0;
    // A fragment of partial-sum pixel m = block * 32 + fr for kernel row ky: halo pixel m + ky * 38.  The swizzle term
    // depends on the row modulo 16 only, and a block shifts the row by a multiple of 32: one address set serves every block.
    // (MX: recomputed per tile from an opaque copy of the lane id -- with the fp4 words parked in registers the kernel sits at its 256-register
    // budget, and every per-lane loop invariant the compiler hoists out of the persistent tile loop is one more spilled register)
    int a_base[KT];
    auto set_a_base = [&](int fr_, int fh_) {
#pragma unroll
        for (int ky = 0; ky < KT; ++ky) {
            const int row = wrot * 32 + fr_ + ky * HWD;
            a_base[ky] = row * 128 + ((fh_ ^ ((row >> 1) & 7)) << 4);
        }
    };
    if (!STREAM) set_a_base(fr, fh);
    // (scale, shift) of the folded InstanceNorm for 16-byte channel group tid & 7 of the current image
    const int c8 = tid & 7;

    for (;;) {
        const int n = tile / tpi, r = tile - n * tpi;
        const int y0 = (r / tiles_x) * PH, x0 = (r % tiles_x) * PW;
        float4 nv[4];                                    // (mean, rstd) x 2 channels, x 4: loaded BEFORE the DMA is issued, so
        if (d.in_norm) {                                 // that waiting for them does not wait for the halo
#pragma unroll
            for (int k = 0; k < 4; ++k) nv[k] = *(const float4*)(d.in_norm + ((long)n * 64 + c8 * 8) * 2 + k * 4);
        }
        unsigned qq[MX ? ROUNDS : 1][2];              // (MX) fp4 words of the residual / the rounded value of the thread's pieces, 8 channels each
        auto stage_f32 = [&](auto lo_plane) {
            constexpr bool LO = decltype(lo_plane)::value;     // (X3) the second plane: fp16((p - fp16(p)) * 2^11)
            // ---- halo through registers: thread t handles 8-channel group t & 7 of pixels (t >> 3) + 32 j
            const float* __restrict__ inf = (const float*)d.in;
            float sc[8], sh[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sc[2 * k] = d.in_norm ? nv[k].y : 1.f; sh[2 * k] = d.in_norm ? -nv[k].x * nv[k].y : 0.f;
                sc[2 * k + 1] = d.in_norm ? nv[k].w : 1.f; sh[2 * k + 1] = d.in_norm ? -nv[k].z * nv[k].w : 0.f;
            }
            const float lo = (d.in_norm && d.in_relu) ? 0.f : -3.0e38f;
            int t8 = tid >> 3;
            asm volatile("" : "+v"(t8));
            const int wbase = t8 * 128 + ((c8 ^ ((t8 >> 1) & 7)) << 4);
            // rolling pipeline: round j's two 16-byte loads are issued RING rounds before they are converted and written to LDS, so the
            // memory latency is paid once per tile (in batches of four rounds it was paid five times: 17 us per tile, 0.55 ms per launch)
            constexpr int RING = 4;
            float4 ring[RING][2];
            unsigned okmask = 0;
            const float lo_scale = __builtin_ldexpf(1.f, -d.c_lo_exp), hi_scale = __builtin_ldexpf(1.f, d.c_hi_exp);   // the converts divide by their scale
            auto issue = [&](int j, float4 (&rg)[2]) {
                const int hp = t8 + j * (NT / 8);
                const int hy = (hp * 1725) >> 16, hx = hp - hy * HWD;
                const int iy = y0 - 3 + hy, ix = x0 - 3 + hx;
                int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
                int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
                ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);
                const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
                if ((hp < HPIX) & (inb | refl)) okmask |= 1u << j;
                const float* src = inf + ((size_t)((n * d.H + ry) * d.W + rx) * 64 + c8 * 8);
                rg[0] = *(const float4*)src; rg[1] = *(const float4*)(src + 4);
            };
            auto finish = [&](int j, const float4 (&rg)[2]) {
                const bool ok = (okmask >> j) & 1u;
                const float a[8] = {rg[0].x, rg[0].y, rg[0].z, rg[0].w, rg[1].x, rg[1].y, rg[1].z, rg[1].w};
                u32x4 o;
                unsigned qlo = 0, qhi = 0;
                // per channel pair k: normalise, round to fp16 (two per instruction); (MX) residuals p - fp16(p) in one v_fma_mix each, both
                // planes to fp4 by the scaled converts (the byte select must be a literal: hence the macro; see conv3x3_halo_c.hip)
#define GDT_H7_PAIR(k)                                                                                                                     \
                {                                                                                                                          \
                    const float p0 = fmaxf(fmaf(a[2 * k], sc[2 * k], sh[2 * k]), lo), p1 = fmaxf(fmaf(a[2 * k + 1], sc[2 * k + 1], sh[2 * k + 1]), lo); \
                    unsigned w;                                                                                                            \
                    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(p0), "v"(p1));                                                       \
                    if (MX) {                                                                                                              \
                        float l0, l1;                                                                                                      \
                        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(w), "v"(p0));                \
                        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(w), "v"(p1));                \
                        qlo = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(qlo, l0, l1, lo_scale, k);                                          \
                        qhi = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(qhi, __builtin_bit_cast(f16x2, w), hi_scale, k);                     \
                    }                                                                                                                      \
                    if (X3 && LO) {                                                                                                        \
                        float l0, l1;                                                                                                      \
                        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(w), "v"(p0));                \
                        asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(w), "v"(p1));                \
                        l0 *= 2048.f; l1 *= 2048.f;                                                                                         \
                        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(l0), "v"(l1));                                                   \
                    }                                                                                                                      \
                    o[k] = ok ? w : 0u;                                                                                                    \
                }
                GDT_H7_PAIR(0) GDT_H7_PAIR(1) GDT_H7_PAIR(2) GDT_H7_PAIR(3)
#undef GDT_H7_PAIR
                if (MX) { qq[MX ? j : 0][0] = ok ? qlo : 0u; qq[MX ? j : 0][1] = ok ? qhi : 0u; }
                // row hp = t8 + 32 j: the swizzle term (hp >> 1) & 7 does not depend on j, so the 17 addresses are one base + j * 4 KB
                // (ROUNDS * 32 <= LROWS: every row exists)
                *(u32x4*)(smem + wbase + j * (NT / 8) * 128) = o;
            };
#pragma unroll
            for (int j = 0; j < RING; ++j) issue(j, ring[j]);
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) {
                finish(j, ring[j % RING]);
                if (j + RING < ROUNDS) issue(j + RING, ring[j % RING]);
                if (MX) __builtin_amdgcn_sched_barrier(0);        // (round by round: interleaved across rounds the 17 unrolled rounds spill)
            }
            lds_barrier();
        };
        if (F32IN) {
            stage_f32(std::false_type{});
        } else {
        // ---- halo: 67 wave-wide 1 KB DMA rounds (reflect / zero padding resolved in the source address)
        if (!(d.dbg & 8) || tile == span_lo + slot) {
#pragma unroll 1
            for (int j = 0; j < (LOAD_ROWS8 + NWAVE - 1) / NWAVE; ++j) {
                const int row8 = min(wave + j * NWAVE, LOAD_ROWS8 - 1);     // (the surplus round repeats the last one)
                const int hp = row8 * 8 + (lane >> 3);
                const int hy = (hp * 1725) >> 16, hx = hp - hy * HWD;       // hp / 38 for hp < 2^11
                const int iy = y0 - 3 + hy, ix = x0 - 3 + hx;
                int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
                int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
                ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);
                const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
                const bool ok = (hp < HPIX) & (inb | refl);
                const int q = (lane & 7) ^ ((hp >> 1) & 7);                   // XOR swizzle on the 16-byte chunk (as conv_igemm.hip)
                const f16* src = d.in + ((long)((n * d.H + ry) * d.W + rx) * 64 + q * 8);
                glds16(ok ? src : d.zeros, smem + row8 * 1024);
            }
        }
        __syncthreads();                               // DMA landed (the barrier drains vmcnt) and visible

        if (d.in_norm && !(d.dbg & 1)) {
            // The producer's InstanceNorm + ReLU (p2p_networks.py:429-431, the up-sampling block before the head) applied in
            // place to the staged halo.  Thread t always handles 16-byte channel group t & 7 (its eight (scale, shift) pairs
            // are fetched once per tile); pieces of padding pixels stay zero.
            const unsigned lo2 = d.in_relu ? 0u : 0xfbfffbffu;                   // packed fp16 floor: 0 or -65504
            float sc[8], sh[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { sc[2 * k] = nv[k].y; sh[2 * k] = -nv[k].x * nv[k].y; sc[2 * k + 1] = nv[k].w; sh[2 * k + 1] = -nv[k].z * nv[k].w; }
            int t8 = tid >> 3;
            asm volatile("" : "+v"(t8));             // (keeps the per-round pixel arithmetic inside the tile loop: hoisted, it spills)
#pragma unroll 2
            for (int j = 0; j < (HPIX + NT / 8 - 1) / (NT / 8); ++j) {
                const int hp = t8 + j * (NT / 8);
                const int hy = (hp * 1725) >> 16, hx = hp - hy * HWD;
                const int iy = y0 - 3 + hy, ix = x0 - 3 + hx;
                const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
                const bool ok = (hp < HPIX) & (inb | refl);
                char* pp = smem + min(hp, LROWS - 1) * 128 + ((c8 ^ ((hp >> 1) & 7)) << 4);
                const u32x4 raw = *(const u32x4*)pp;
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned w;
                    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(w) : "v"(raw[k]), "v"(sc[2 * k]), "v"(sh[2 * k]));
                    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(w) : "v"(raw[k]), "v"(sc[2 * k + 1]), "v"(sh[2 * k + 1]));
                    asm("v_pk_max_f16 %0, %1, %2" : "=v"(w) : "v"(w), "v"(lo2));
                    o[k] = ok ? w : 0u;
                }
                if (hp < LROWS) *(u32x4*)pp = o;
            }
            lds_barrier();
        }

        }

        // ---- GEMM over the kernel rows: 2 or 3 independent accumulator chains per wave
        int lane_t = lane;
        if (STREAM) { asm volatile("" : "+v"(lane_t)); set_a_base(lane_t & 31, lane_t >> 5); }
        const int fr_t = lane_t & 31, fh_t = lane_t >> 5;
        const int b2 = nb == 3 ? 2 : 1;
        f32x16 acc[3], accl[X3 ? 3 : 1];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[b][e] = 0.f; if (X3) accl[X3 ? b : 0][e] = 0.f; }
        constexpr int PF = 3;                                  // fragment sets in flight (LDS latency vs 3 MFMAs per set)
        // (MX form: every wave issues three MFMA chains -- the waves that own two blocks repeat their second one and drop the result: with
        // the branches of the two-or-three form around 28 + 14 unrolled steps the register allocator spilled ~230 registers)
        auto frags = [&](int ks, f16x8 (&f)[3]) {
            const int off = a_base[ks >> 2] ^ ((ks & 3) << 5);
            f[0] = *(const f16x8*)(smem + off);
            f[1] = *(const f16x8*)(smem + off + NWAVE * 32 * 128);
            if (MX) f[2] = *(const f16x8*)(smem + off + b2 * (NWAVE * 32 * 128));          // (b2 = 1 for the waves with two blocks: a duplicate, discarded)
            else if (nb == 3) f[2] = *(const f16x8*)(smem + off + 2 * NWAVE * 32 * 128);
        };
        if (!(d.dbg & 2)) {
            f16x8 afr[PF][3];
#pragma unroll
            for (int p = 0; p < PF - 1; ++p) frags(p, afr[p]);
            if (STREAM) {
                lane_offsets();
#pragma unroll
                for (int p = 0; p < WRING - 1; ++p) { load_w(p); if (X3) load_wl(p); }
            }
#pragma unroll
            for (int ks = 0; ks < 4 * KT; ++ks) {
                if (ks + PF - 1 < 4 * KT) frags(ks + PF - 1, afr[(ks + PF - 1) % PF]);
                if (STREAM && ks + WRING - 1 < 4 * KT) { load_w(ks + WRING - 1); if (X3) load_wl(ks + WRING - 1); }
                const f16x8 wk = STREAM ? wring[ks % WRING] : bw[ks % NBW];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks % PF][0], wk, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks % PF][1], wk, acc[1], 0, 0, 0);
                if (MX || nb == 3) acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks % PF][2], wk, acc[2], 0, 0, 0);
                if (X3) {                       // a_hi w_lo
                    const f16x8 wl = wring_lo[X3 ? ks % WRING : 0];
#pragma unroll
                    for (int b = 0; b < 2; ++b) accl[X3 ? b : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks % PF][b], wl, accl[X3 ? b : 0], 0, 0, 0);
                    if (nb == 3) accl[X3 ? 2 : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks % PF][2], wl, accl[X3 ? 2 : 0], 0, 0, 0);
                }
            }
        }
        lds_barrier();                                 // the halo has been consumed by every wave
        if constexpr (MX) {
            // ---- correction pass: the fp4 plane over the consumed halo -- 64-byte rows [lo 0-31 | hi 0-31 | lo 32-63 | hi 32-63], 16-byte
            // pieces XOR-swizzled by (row >> 2) & 3 (conflict-free 16-lane fragment reads) -- then 14 MX MFMAs per block, weight fragments
            // (conv3x3_halo_c.hip layout: 16 + 8 bytes of e2m3 values and one E8M0 scale dword per lane) streamed L2 -> registers one ahead
            int t8 = tid >> 3;
            asm volatile("" : "+v"(t8));
            // row hp = t8 + 32 j, piece (2 * (c8 >> 2) + {lo 0, hi 1}) ^ ((hp >> 2) & 3): again independent of j -> one base + j * 2 KB
            const int qbase = t8 * 64 + ((((c8 >> 2) << 1) ^ ((t8 >> 2) & 3)) << 4) + ((c8 & 3) << 2);
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) {
                *(unsigned*)(smem + qbase + j * (NT / 8) * 64) = qq[j][0];
                *(unsigned*)(smem + (qbase ^ 16) + j * (NT / 8) * 64) = qq[j][1];
            }
            typedef int v4i __attribute__((ext_vector_type(4)));
            typedef int v2i __attribute__((ext_vector_type(2)));
            typedef int v6i __attribute__((ext_vector_type(6)));
            typedef int v8i __attribute__((ext_vector_type(8)));
            v6i wq[2]; int wqs[2];
            auto load_wq = [&](int slot, int ms) {
                const v4i qa = *(const v4i*)((const char*)d.wmx_a + ms * 4096 + lo16);
                const v2i qb = *(const v2i*)((const char*)d.wmx_b + ms * 2048 + lo8);
                wq[slot] = __builtin_shufflevector(__builtin_shufflevector(qa, qa, 0, 1, 2, 3, -1, -1), __builtin_shufflevector(qb, qb, 0, 1, -1, -1, -1, -1), 0, 1, 2, 3, 6, 7);
                wqs[slot] = *(const int*)((const char*)d.wmx_s + ms * 1024 + lo4);
            };
            load_wq(0, 0);
            lds_barrier();
            const int a_scale = fh_t ? 127 + d.c_hi_exp : 127 - d.c_lo_exp;       // lanes 0-31 carry a_lo * 2^c_lo_exp, lanes 32-63 a_hi * 2^-c_hi_exp
            const int qrow0 = wrot * 32 + fr_t;
#pragma unroll
            for (int ms = 0; ms < 2 * KT; ++ms) {
                if (ms + 1 < 2 * KT) load_wq((ms + 1) & 1, ms + 1);
                const int row = qrow0 + (ms >> 1) * HWD;
                const int off = row * 64 + (((((ms & 1) << 1) | fh_t) ^ ((row >> 2) & 3)) << 4);
                const v8i wv = __builtin_shufflevector(wq[ms & 1], wq[ms & 1], 0, 1, 2, 3, 4, 5, -1, -1);
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                        const v4i av4 = *(const v4i*)(smem + off + (b == 2 ? b2 : b) * (NWAVE * 32 * 64));
                        const v8i av = __builtin_shufflevector(av4, av4, 0, 1, 2, 3, -1, -1, -1, -1);
                        acc[b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, wv, acc[b], 4, 2, 0, a_scale, 0, wqs[ms & 1]);
                    }
            }
            lds_barrier();                             // the fp4 plane has been consumed
        }
        if constexpr (X3) {
            // ---- second plane: a_lo over the consumed halo (read again, split again), then a_lo w_hi into the correction accumulators
            stage_f32(std::true_type{});
            int lane_u = lane;
            asm volatile("" : "+v"(lane_u));
            set_a_base(lane_u & 31, lane_u >> 5);
            f16x8 afr[PF][3];
#pragma unroll
            for (int p = 0; p < PF - 1; ++p) frags(p, afr[p]);
            lane_offsets();
#pragma unroll
            for (int p = 0; p < WRING - 1; ++p) load_w(p);
#pragma unroll
            for (int ks = 0; ks < 4 * KT; ++ks) {
                if (ks + PF - 1 < 4 * KT) frags(ks + PF - 1, afr[(ks + PF - 1) % PF]);
                if (ks + WRING - 1 < 4 * KT) load_w(ks + WRING - 1);
                const f16x8 wk = wring[ks % WRING];
#pragma unroll
                for (int b = 0; b < 2; ++b) accl[X3 ? b : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks % PF][b], wk, accl[X3 ? b : 0], 0, 0, 0);
                if (nb == 3) accl[X3 ? 2 : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[ks % PF][2], wk, accl[X3 ? 2 : 0], 0, 0, 0);
            }
            lds_barrier();                             // the lo plane has been consumed
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][e] += accl[X3 ? b : 0][e] * (1.f / 2048.f);
        }
        float* P = (float*)smem;
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (b < nb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) P[((wrot + b * NWAVE) * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh_t) * PSTRIDE + fr_t] = acc[b][e];
            }
        lds_barrier();
        for (int idx = tid; idx < cout * PH * PW && !(d.dbg & 4); idx += NT) {
            const int co = idx >> 8, y = (idx >> 5) & 7, x = idx & 31;
            float v = d.bias ? d.bias[co] : 0.f;
#pragma unroll
            for (int kx = 0; kx < KT; ++kx) v += P[(y * HWD + x + kx) * PSTRIDE + kx * cout + co];
            if (d.act == 1) v = tanhf(v);
            else if (d.act == 2) v = 1.f / (1.f + __expf(-v));
            if (y0 + y < d.H && x0 + x < d.W) d.out_f32[(((long)n * cout + co) * d.H + y0 + y) * d.W + x0 + x] = v;
        }
        tile += S;
        if (tile >= span_hi) break;
        lds_barrier();                                 // the partial sums have been read: the buffer may be refilled
    }
}

}  // namespace

// 7x7 stride-1 pad-3 conv, 64 input channels, <= 4 output channels, fp32 NCHW output, fragment-ordered row-split weights
bool gdt_conv_head7_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_HEAD7"); return e ? atoi(e) : 1; }();
    return mode != 0 && d.w_frag && d.out_f32 && d.Cin == 64 && d.Cout >= 1 && d.Cout <= 4 && d.ntaps == KT &&
           (long)d.N * d.H * d.W * 64 < (1L << 31);
}

int gdt_launch_conv_head7(const ConvLaunch& d, hipStream_t stream) {
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    constexpr int lds = HBYTES;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_head7_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_head7_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_head7_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_head7_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    const int ntiles = d.N * ((d.W + PW - 1) / PW) * ((d.H + PH - 1) / PH);
    static const int wgs = [] { const char* e = getenv("GDT_HEAD7_WGS"); return e ? atoi(e) : 2; }();
    const int grid = min(wgs * cus, (ntiles + 7) / 8 * 8);     // two workgroups per CU: one loads while the other computes
    static const int dbg = [] { const char* e = getenv("GDT_HEAD7_DBG"); return e ? atoi(e) : 0; }();
    ConvLaunch dd = d;
    dd.dbg = dbg;
    if (d.in_f32 && d.w_frag2) hipLaunchKernelGGL((conv_head7_kernel<true, false, true>), dim3(grid), dim3(NT), lds, stream, dd, ntiles);       // f16x3: w_frag2 = the lo parts
    else if (d.in_f32 && d.wmx_a && d.wmx_b && d.wmx_s) hipLaunchKernelGGL((conv_head7_kernel<true, true>), dim3(grid), dim3(NT), lds, stream, dd, ntiles);
    else if (d.in_f32) hipLaunchKernelGGL(conv_head7_kernel<true>, dim3(grid), dim3(NT), lds, stream, dd, ntiles);
    else hipLaunchKernelGGL(conv_head7_kernel<false>, dim3(grid), dim3(NT), lds, stream, dd, ntiles);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}
