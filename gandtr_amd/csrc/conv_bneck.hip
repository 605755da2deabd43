// Whole ResNet Bottleneck (identity form) in ONE launch, fp16 mode:
//     y = ReLU( x + BN3(W_e * ReLU( BN2( W_3 (*) ReLU( BN1( W_r * x ))))))        torchvision Bottleneck as restated in oracle/gandtr_oracle.py:116-133
// with W_r: 1x1 C -> MID, W_3: 3x3 / stride 1 / zero pad 1 MID -> MID, W_e: 1x1 MID -> C, BatchNorm(eval) folded into the weights /
// biases at pack time.  ResNet-101 layer1 (C 256, MID 64, 256 x 256 maps at a 1024^2 input) and layer2 (C 512, MID 128, 128 x 128).
//
// Why: these layers are HBM-bound on their block-boundary tensors (1.07 GB / 0.54 GB per batch-32 tensor).  Layer by layer a block moves
// x (reduce in) + r + r + t + t + x (residual) + y = 4.3 GB in layer1 (2.0 GB in layer2) and takes 1.0 ms (0.58 ms); fused, r and t never
// leave the CU: x in (+ halo), y out = 2.4 GB (1.1 GB).
//
// One persistent 512-thread workgroup per CU walks XCD-chunked PH x 16 output patches:
//   reduce   the (PH + 2) x 18 input halo goes global -> registers -> LDS in 64-channel chunks: LDS holds chunk c, registers chunk c + 1,
//            the loads of chunk c + 2 are in flight (the first two chunks of the NEXT patch are requested during the expand phase);
//            r = ReLU(W_r x + b_r) for every halo pixel accumulates in registers (the 1x1 conv is recomputed on the halo ring: 1.27x /
//            1.41x its FLOPs, a quarter of the block's), is zeroed outside the image (the 3x3 conv pads r, not x) and written to LDS
//            as fp16 over the x buffers;
//   3x3      t = ReLU(W_3 (*) r + b_3) from the LDS-resident r halo.  W_3 (72 / 288 KB) does not fit beside the tensors and every wave
//            needs half / a quarter of it per patch: streamed per wave from L2 it made the phase 44 k cycles of exposed latency, and
//            the L2 -> L1 path (~33 B/clk per CU) carried 8 x the bytes.  So the workgroup stages it once, (tap, 64-channel plane)
//            piece by piece, through the same registers -> LDS pipeline and every wave reads fragments from LDS;
//   expand   y = ReLU(W_e t + b_e + x): a wave owns ONE 64-channel pair of the output (its 2 x MID/16 weight fragments stay in
//            registers for the whole launch) and walks pixel blocks; every 32 x 64 block is transposed through a 4 KB patch of the
//            wave's own so that lanes store -- and fetch the residual x as -- whole 128-byte lines.
//   No LDS-DMA: with a `global_load_lds` in flight the compiler turns every later vector-memory wait into vmcnt(0), which serialises the
//   weight loads behind the HBM stream; through registers every wait is counted exactly.  Within a step the (L2-hit) weight loads are
//   issued BEFORE the (HBM) activation loads: vector-memory operations retire in issue order, and a wait for a weight fragment must
//   not have to sit out a younger HBM miss.
//   MFMA operands are swapped (D = W A^T): a lane holds 4 consecutive channels of one pixel.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gdt_common.h"

namespace {

constexpr int NT = 512, NWAVE = 8, PW = 16, HW_ = PW + 2;

// bias of the lane's 4 channels base + 8 g + 4 fh + {0..3}: both candidates come through UNIFORM addresses (scalar loads: their own counter,
// never queued behind the HBM loads the vector-memory counter is full of), the lane picks by fh
__device__ __forceinline__ float4 bias4(const float* __restrict__ b, int base_uniform, int g, int fh) {
    const float4 lo = *(const float4*)(b + base_uniform + 8 * g), hi = *(const float4*)(b + base_uniform + 8 * g + 4);
    return fh ? hi : lo;
}

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// DS: the block's shortcut is a 1x1 projection W_d (CIN -> C, BatchNorm folded) of x instead of x itself -- layer1's first block (CIN 64): the x halo
// chunk stays in LDS through the expand phase, which accumulates W_d x into the same accumulators as W_e t (K = MID + CIN), r and t get regions of their own.
template <int C, int MID, int PH, int CIN = C, bool DS = false>
struct Geo {
    static constexpr int HH = PH + 2, HPIX = HH * HW_, HB = (HPIX + 31) / 32, HPAD = HB * 32;     // halo pixels, 32-pixel blocks
    static constexpr int P = PH * PW, PB = P / 32;                 // output pixels, 32-pixel blocks (a block = two patch rows)
    static constexpr int NC = CIN / 64, MB = MID / 32, MP = MID / 64;                                 // x chunks, mid 32-blocks, mid 64-planes
    static constexpr int XBYTES = HPAD * 128;                      // one x chunk buffer (64 channels x fp16 per halo pixel)
    static constexpr int NPX = (XBYTES / 16 + NT - 1) / NT;        // 16-byte units of a chunk per thread
    static constexpr int RBYTES = MP * HPAD * 128;                 // r halo, MP planes of 128-byte rows: over the x buffers
    static constexpr int TBYTES = MP * P * 128;                    // t: over x buffer 1 where r leaves it free (layer1), else its own region
    static constexpr bool T_OVER_X = !DS && RBYTES + TBYTES <= 2 * XBYTES;
    static constexpr int W3BYTES = 9 * MID * MID * 2;              // W_3: resident in LDS when it fits beside the two x buffers (layer1: to the byte) ...
    static constexpr bool W3_RES = !DS && 2 * XBYTES + W3BYTES <= 160 * 1024 && RBYTES + TBYTES <= 2 * XBYTES;
    static constexpr int W3P = MID * MID * 2, NW3 = 9;             // ... else staged tap by tap, two buffers
    static constexpr int NPW = W3_RES ? 1 : W3P / 16 / NT;         // 16-byte units of a tap per thread
    static constexpr int R_OFF = DS ? XBYTES : 0;                  // (DS: x buffer 0 keeps the chunk, r lives in buffer 1's place)
    static constexpr int T_OFF = T_OVER_X ? RBYTES : 2 * XBYTES;
    static constexpr int W3_OFF = 2 * XBYTES + (T_OVER_X ? 0 : TBYTES);
    static constexpr int LDS = W3_OFF + (W3_RES ? W3BYTES : 2 * W3P);
    static constexpr int ECP = C / 64, EPW = (ECP * PB) / NWAVE;   // expand: 64-channel pairs; (pair, pixel block) tiles per wave
    static_assert((!DS || (NC == 1 && RBYTES <= XBYTES)) && RBYTES <= 2 * XBYTES && NWAVE * 4096 <= RBYTES && LDS <= 160 * 1024 && (W3_RES || W3P % (16 * NT) == 0), "LDS layout");
    static_assert((NWAVE / MB) * 3 >= HB, "reduce tiles: 3 pixel blocks per wave");
    static_assert(PB * MB == 2 * NWAVE, "3x3 tiles: two per wave");
    static_assert(ECP <= NWAVE ? (NWAVE % ECP == 0 && PB % (NWAVE / ECP) == 0) : false, "expand tiles");
};

struct BneckLaunch {
    unsigned long long* stamps;       // diagnostic builds only (-DGDT_BNECK_STAMP): per-wave s_memtime totals per phase
    const f16* x; f16* y;
    const f16* wr; const f16* w3; const f16* we;         // fragment order [cout/32][K/16][64 lanes][8 halves]
    const float* br; const float* b3; const float* be;   // folded BatchNorm shifts
    const f16* wd; const float* bd;                      // projection shortcut (DS form), else null
    int N, H, W;
};

// RAGGED: maps the patches do not tile exactly (real images: H, W arbitrary) -- the last patch row / column hangs over the image: its x loads are clamped, r is
// zeroed there (as at every image border), residual fetches are clamped and the stores of pixels outside the image are masked off
// several independent geometries (the levels of a pyramid) in one launch: gdt_common.h MultiConv, here with this kernel's own descriptor
struct MultiBneck {
    int nlev;
    int prefix[GDT_MAX_LEVELS + 1];
    int ntiles[GDT_MAX_LEVELS];
    BneckLaunch lev[GDT_MAX_LEVELS];
};

// (the body takes its workgroup index and grid size as arguments: the multi-geometry entry runs it per level)
template <int C, int MID, int PH, int CIN = C, bool DS = false, bool RAGGED = false>
__device__ __forceinline__ void conv_bneck_body(const BneckLaunch& d, const int ntiles, const int bid, const int gdim) {
    using G = Geo<C, MID, PH, CIN, DS>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbuf0 = smem;
    char* const xbuf1 = smem + G::XBYTES;
    char* const rbuf = smem + G::R_OFF;            // (over the x buffers, once the reduce phase has consumed them; DS: beside the kept chunk)
    char* const tbuf = smem + G::T_OFF;
    char* const w3s = smem + G::W3_OFF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int tiles_x = (d.W + PW - 1) / PW, tiles_y = (d.H + PH - 1) / PH, tpi = tiles_x * tiles_y;

    // XCD-chunked persistent schedule (conv_head7.hip): the CUs of one XCD hold neighbouring patches, whose halos overlap in its L2
    const int per_xcd = (ntiles + 7) >> 3, S = gdim >> 3;
    const int xcd = bid & 7, slot = bid >> 3;
    const int span_lo = xcd * per_xcd, span_hi = min(span_lo + per_xcd, ntiles);
    int tile = span_lo + slot;
    if (tile >= span_hi) return;

    // ---- x chunk, global -> registers -> LDS.  Unit e = tid + k * NT: halo pixel e >> 3, 16-byte source piece e & 7, stored at piece
    // position (e & 7) ^ swizzle(pixel) of its 128-byte LDS row.  (Pixels outside the image: any valid address, r is zeroed there.)
    struct XRegs { f16x8 v[G::NPX]; };
    auto load_x = [&](int tl, int c, XRegs& xr) {
        const int n = tl / tpi, r = tl - n * tpi;
        const int y0 = (r / tiles_x) * PH, x0 = (r % tiles_x) * PW;
        int t = tid;
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int k = 0; k < G::NPX; ++k) {
            const int e = min(t + k * NT, G::HPAD * 8 - 1);
            const int hp = min(e >> 3, G::HPIX - 1);
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int iy = min(max(y0 - 1 + hy, 0), d.H - 1), ix = min(max(x0 - 1 + hx, 0), d.W - 1);
            xr.v[k] = *(const f16x8*)(d.x + ((size_t)((n * d.H + iy) * d.W + ix) * CIN + c * 64 + (e & 7) * 8));
        }
    };
    auto store_x = [&](const XRegs& xr, char* dst) {
        int t = tid;
        asm volatile("" : "+v"(t));
#pragma unroll
        for (int k = 0; k < G::NPX; ++k) {
            const int e = t + k * NT;
            const int row = e >> 3;
            if (e < G::HPAD * 8) *(f16x8*)(dst + row * 128 + (((e & 7) ^ ((row >> 1) & 7)) << 4)) = xr.v[k];
        }
    };
    // ---- W_3 (staged form): tap q = its MB * (MID / 16) fragments (mid-out block cb, k-group kg), 1 KB each, copied linearly [cb][kg][64 lanes][16 B]
    struct WRegs { f16x8 v[G::NPW]; };
    constexpr int KS3 = 9 * (MID / 16), KT3 = MID / 16;
    auto load_w3 = [&](int q, WRegs& wq) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int k = 0; k < G::NPW; ++k) {
            const int u = tt + k * NT, f = u >> 6;               // fragment f = cb * KT3 + kg
            wq.v[k] = *(const f16x8*)((const char*)d.w3 + (((f / KT3) * KS3 + q * KT3 + f % KT3) << 10) + (u & 63) * 16);
        }
    };
    auto store_w3 = [&](const WRegs& wq, char* dst) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int k = 0; k < G::NPW; ++k) *(f16x8*)(dst + (tt + k * NT) * 16) = wq.v[k];
    };

    // ---- per-wave tile assignments
    // reduce: mid block rmb, halo pixel blocks rpb + RSTEP * j (j = 0..2): one weight fragment feeds three MFMAs
    constexpr int RSTEP = NWAVE / G::MB;
    const int rmb = wave % G::MB, rpb = wave / G::MB;
    // 3x3: mid-out block cmb, pixel blocks cpb, cpb + CPS
    constexpr int CPS = G::PB / 2;
    const int cmb = wave % G::MB, cpb = wave / G::MB;
    // expand: channel pair ecp, pixel blocks epb0 + j (j < EPW)
    const int ecp = wave % G::ECP, epb0 = (wave / G::ECP) * G::EPW;

    // ---- once per launch: W_3 into LDS (layer1: source layout [cb][k-group][64 lanes][16 B])
    if (G::W3_RES) {
        for (int u = tid; u < G::W3BYTES / 16; u += NT) *(f16x8*)(w3s + u * 16) = *(const f16x8*)((const char*)d.w3 + u * 16);
    }

    XRegs xa, xb;                  // chunk c + 1 / c + 2 of the stream (roles alternate)
    load_x(tile, 0, xa);
    if (G::NC > 1) load_x(tile, 1, xb);
    store_x(xa, xbuf0);
    lds_barrier();

#ifdef GDT_BNECK_STAMP
    unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[4] = {0, 0, 0, 0};
#define BN_STAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_t; st_t = now_; }
#else
#define BN_STAMP(k)
#endif
    for (;;) {
        // (uniform base + 32-bit lane offset, refreshed per patch behind an opaque copy: as 64-bit per-lane addresses the fragment
        // addresses are loop invariants that the compiler hoists out of the patch loop and spills)
        unsigned lo16 = lane * 16;
        asm volatile("" : "+v"(lo16));
        const int n = tile / tpi, rr = tile - n * tpi;
        const int y0 = (rr / tiles_x) * PH, x0 = (rr % tiles_x) * PW;
        const int next = tile + S;
        const bool has_next = next < span_hi;

        // ================================================================ reduce: r = ReLU(W_r x + b_r) on the halo
        // invariant at the top of iteration c: LDS buffer c & 1 holds chunk c (visible to all), registers (c odd ? xa : xb) hold chunk c + 1
        f32x16 racc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) racc[j][e] = 0.f;
        WRegs wq0, wq1;                                              // W_3 pieces 0 and 1: requested under the last two chunks
        f16x8 wrn[4];                                                // (W_r from L2: the fragments of the NEXT chunk, requested one chunk ahead)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wrn[ks] = *(const f16x8*)((const char*)d.wr + ((rmb * (CIN / 16) + ks) << 10) + lo16);
#pragma unroll
        for (int c = 0; c < G::NC; ++c) {
            XRegs& xin = (c & 1) ? xa : xb;                          // holds chunk c + 1
            XRegs& xld = (c & 1) ? xb : xa;                          // free: receives chunk c + 2
            f16x8 wrc[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wrc[ks] = wrn[ks];
            if (c + 1 < G::NC) {                                     // weight loads first: older than this iteration's HBM loads
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wrn[ks] = *(const f16x8*)((const char*)d.wr + ((rmb * (CIN / 16) + (c + 1) * 4 + ks) << 10) + lo16);
            }
            if (!G::W3_RES && (c == G::NC - 2 || G::NC == 1)) load_w3(0, wq0);
            if (!G::W3_RES && c == G::NC - 1) load_w3(1, wq1);
            if (c + 2 < G::NC) load_x(tile, c + 2, xld);
            const char* xbc = (c & 1) ? xbuf1 : xbuf0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int pb = rpb + RSTEP * j;
                    if (pb < G::HB) {
                        const int row = pb * 32 + fr;
                        const f16x8 af = *(const f16x8*)(xbc + row * 128 + (((2 * ks + fh) ^ ((row >> 1) & 7)) << 4));
                        racc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wrc[ks], af, racc[j], 0, 0, 0);
                    }
                }
            }
            if (c + 1 < G::NC) store_x(xin, (c & 1) ? xbuf0 : xbuf1);   // chunk c + 1 -> the buffer chunk c - 1 has left (barrier of iteration c - 1)
            lds_barrier();
        }
        // r -> LDS (fp16) over the x buffers (all consumed: the loop ends with a barrier): lane holds pixel `row`, channels
        // rmb * 32 + 8 g + 4 fh + {0..3}; zero outside the image
        {
            float4 brv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) brv[g] = bias4(d.br, rmb * 32, g, fh);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int pb = rpb + RSTEP * j;
                if (pb < G::HB) {
                    const int row = pb * 32 + fr;
                    const int hy = row / HW_, hx = row - hy * HW_;
                    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                    const bool inside = (row < G::HPIX) & ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
                    char* rp = rbuf + (rmb >> 1) * (G::HPAD * 128) + row * 128;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b = brv[g];
                        f16x4 v;
                        v[0] = (f16)(inside ? fmaxf(racc[j][4 * g] + b.x, 0.f) : 0.f);
                        v[1] = (f16)(inside ? fmaxf(racc[j][4 * g + 1] + b.y, 0.f) : 0.f);
                        v[2] = (f16)(inside ? fmaxf(racc[j][4 * g + 2] + b.z, 0.f) : 0.f);
                        v[3] = (f16)(inside ? fmaxf(racc[j][4 * g + 3] + b.w, 0.f) : 0.f);
                        *(f16x4*)(rp + ((((rmb & 1) * 4 + g) ^ ((row >> 1) & 7)) << 4) + fh * 8) = v;
                    }
                }
            }
        }
        if (!G::W3_RES) store_w3(wq0, w3s);
        lds_barrier();                     // r (and tap 0 of W_3) complete
        BN_STAMP(0)
        // the first two x chunks of the NEXT patch and the residual rows of this one are requested now: the memory system works through
        // the 3x3 phase, which itself touches LDS only (layer1) / L2 only (layer2's W_3 taps, which queue behind these loads once)
        if (has_next) { load_x(next, 0, xa); if (G::NC > 1) load_x(next, 1, xb); }
        const int pl = lane >> 3, pc = lane & 7;
        auto res_off = [&](int jb, int i) -> size_t {           // row this lane stores / fetches the residual of: output pixel 8 i + (lane >> 3) of
            const int prow = (epb0 + jb) * 32 + 8 * i + pl;     // pixel block epb0 + jb (patch row prow / 16, column prow % 16), 16-byte piece lane & 7
            const int y = y0 + (prow >> 4), x = x0 + (prow & 15);
            if (RAGGED) return ((size_t)((n * d.H + min(y, d.H - 1)) * d.W + min(x, d.W - 1))) * C + ecp * 64 + pc * 8;       // (a valid address; see res_ok)
            return ((size_t)((n * d.H + y) * d.W + x)) * C + ecp * 64 + pc * 8;
        };
        auto res_ok = [&](int jb, int i) -> bool {
            const int prow = (epb0 + jb) * 32 + 8 * i + pl;
            return !RAGGED || ((y0 + (prow >> 4) < d.H) & (x0 + (prow & 15) < d.W));
        };

        // ================================================================ 3x3: t = ReLU(W_3 (*) r + b_3)
        // wave: mid-out block cmb, pixel blocks cpb and cpb + CPS.  Per piece q: LDS buffer q & 1 holds piece q, registers piece q + 1,
        // the loads of piece q + 2 are in flight; one barrier per piece.
        {
            f32x16 tacc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) tacc[j][e] = 0.f;
            // (from an opaque copy of the lane id: the 72 / 144 swizzled fragment addresses of this phase are functions of the lane only, i.e.
            // loop invariants of the persistent patch loop -- hoisted, they are spilled in the prologue and every reload in here is a
            // vector-memory wait that queues behind the HBM loads just requested)
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            const int fr = lane_t & 31, fh = lane_t >> 5;
            int hbase[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) hbase[j] = ((cpb + CPS * j) * 2 + (fr >> 4)) * HW_ + (fr & 15);
            if (G::W3_RES) {
                auto frags = [&](int st, f16x8 (&f)[3]) {            // fragments of k-step st = (tap, plane, k): weights, two pixel blocks
                    const int t = st / KT3, mp = (st % KT3) >> 2, ks = st & 3;
                    f[0] = *(const f16x8*)(w3s + ((cmb * KS3 + st) << 10) + lane * 16);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int hrow = hbase[j] + (t / 3) * HW_ + t % 3;
                        f[1 + j] = *(const f16x8*)(rbuf + mp * (G::HPAD * 128) + hrow * 128 + (((2 * ks + fh) ^ ((hrow >> 1) & 7)) << 4));
                    }
                };
                f16x8 fq[3][3];                                      // three steps in flight (LDS latency vs two MFMAs per step)
                frags(0, fq[0]); frags(1, fq[1]);
#pragma unroll
                for (int st = 0; st < KS3; ++st) {
                    if (st + 2 < KS3) frags(st + 2, fq[(st + 2) % 3]);
                    tacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[st % 3][0], fq[st % 3][1], tacc[0], 0, 0, 0);
                    tacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fq[st % 3][0], fq[st % 3][2], tacc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < G::NW3; ++q) {
                    WRegs& win = (q & 1) ? wq0 : wq1;                    // holds tap q + 1
                    WRegs& wld = (q & 1) ? wq1 : wq0;                    // free: receives tap q + 2
                    if (q + 2 < G::NW3) load_w3(q + 2, wld);
                    const char* wsq = w3s + (q & 1) * G::W3P;
#pragma unroll
                    for (int kg = 0; kg < KT3; ++kg) {
                        const int mp = kg >> 2, ks = kg & 3;
                        const f16x8 wf = *(const f16x8*)(wsq + ((cmb * KT3 + kg) << 10) + lane * 16);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int hrow = hbase[j] + (q / 3) * HW_ + q % 3;
                            const f16x8 af = *(const f16x8*)(rbuf + mp * (G::HPAD * 128) + hrow * 128 + (((2 * ks + fh) ^ ((hrow >> 1) & 7)) << 4));
                            tacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, af, tacc[j], 0, 0, 0);
                        }
                    }
                    if (q + 1 < G::NW3) { store_w3(win, w3s + ((q + 1) & 1) * G::W3P); lds_barrier(); }
                }
            }
            // (t has a region of its own -- behind r in the x buffers, or separate: no barrier before it is written)
            float4 b3v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) b3v[g] = bias4(d.b3, cmb * 32, g, fh);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (cpb + CPS * j) * 32 + fr;
                char* tp = tbuf + (cmb >> 1) * (G::P * 128) + row * 128;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bb = b3v[g];
                    f16x4 v;
                    v[0] = (f16)fmaxf(tacc[j][4 * g] + bb.x, 0.f); v[1] = (f16)fmaxf(tacc[j][4 * g + 1] + bb.y, 0.f);
                    v[2] = (f16)fmaxf(tacc[j][4 * g + 2] + bb.z, 0.f); v[3] = (f16)fmaxf(tacc[j][4 * g + 3] + bb.w, 0.f);
                    *(f16x4*)(tp + ((((cmb & 1) * 4 + g) ^ ((row >> 1) & 7)) << 4) + fh * 8) = v;
                }
            }
        }
        // the wave's W_e fragments (one 64-channel pair: 2 x MID / 16, L2 hits), fetched per patch: kept across the reduce phase they
        // cost 32 - 64 registers there, and its six-chunk register pipeline then spills behind its own HBM loads
        f16x8 wef[2][MID / 16];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kg = 0; kg < MID / 16; ++kg) wef[j][kg] = *(const f16x8*)((const char*)d.we + (((ecp * 2 + j) * (MID / 16) + kg) << 10) + lo16);
        lds_barrier();                     // t complete; r consumed (its buffer becomes the waves' store patches)
        BN_STAMP(1)

        // ================================================================ expand: y = ReLU(W_e t + b_e + x)
        {
            char* patch = rbuf + wave * 4096;
            float4 bev[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) bev[j][g] = bias4(d.be, ecp * 64 + j * 32, g, fh);
            f16x8 res[2][4];                                           // residual rows: block jb in res[jb & 1], block jb + 1 requested meanwhile
            if (!DS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) res[0][i] = *(const f16x8*)(d.x + res_off(0, i));
            }
            f16x8 wdf[2][DS ? CIN / 16 : 1];                           // (DS) the wave's W_d fragments
            if (DS) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int kg = 0; kg < CIN / 16; ++kg) wdf[j][kg] = *(const f16x8*)((const char*)d.wd + (((ecp * 2 + j) * (CIN / 16) + kg) << 10) + lo16);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b2 = bias4(d.bd, ecp * 64 + j * 32, g, fh);
                        bev[j][g].x += b2.x; bev[j][g].y += b2.y; bev[j][g].z += b2.z; bev[j][g].w += b2.w;
                    }
            }
#pragma unroll
            for (int jb = 0; jb < G::EPW; ++jb) {
                const int epb = epb0 + jb;
                size_t goff[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) goff[i] = res_off(jb, i);
                if (!DS && jb + 1 < G::EPW) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) res[(jb + 1) & 1][i] = *(const f16x8*)(d.x + res_off(jb + 1, i));
                }
                const int row = epb * 32 + fr;
                f32x16 eacc[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) eacc[j][e] = 0.f;
#pragma unroll
                for (int kg = 0; kg < MID / 16; ++kg) {
                    const f16x8 tf = *(const f16x8*)(tbuf + (kg >> 2) * (G::P * 128) + row * 128 + (((2 * (kg & 3) + fh) ^ ((row >> 1) & 7)) << 4));
#pragma unroll
                    for (int j = 0; j < 2; ++j) eacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wef[j][kg], tf, eacc[j], 0, 0, 0);
                }
                if (DS) {                    // + W_d x: the pixel's own row of the x halo chunk kept in buffer 0
                    const int hrow = ((row >> 4) + 1) * HW_ + (row & 15) + 1;
#pragma unroll
                    for (int kg = 0; kg < CIN / 16; ++kg) {
                        const f16x8 xf = *(const f16x8*)(xbuf0 + hrow * 128 + (((2 * kg + fh) ^ ((hrow >> 1) & 7)) << 4));
#pragma unroll
                        for (int j = 0; j < 2; ++j) eacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wdf[j][kg], xf, eacc[j], 0, 0, 0);
                    }
                }
                // accumulators (+ bias) -> patch: lane holds pixel fr, channels j * 32 + 8 g + 4 fh + {0..3} of the pair
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bb = bev[j][g];
                        f16x4 v;
                        v[0] = (f16)(eacc[j][4 * g] + bb.x); v[1] = (f16)(eacc[j][4 * g + 1] + bb.y);
                        v[2] = (f16)(eacc[j][4 * g + 2] + bb.z); v[3] = (f16)(eacc[j][4 * g + 3] + bb.w);
                        *(f16x4*)(patch + fr * 128 + (((j * 4 + g) ^ (fr & 7)) << 4) + fh * 8) = v;
                    }
                }
                // (wave-private patch: the wave's own LDS operations are executed in order, no barrier)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pr = 8 * i + pl;
                    const f16x8 v = *(const f16x8*)(patch + pr * 128 + ((pc ^ (pr & 7)) << 4));
                    f16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (f16)fmaxf((float)v[e] + (DS ? 0.f : (float)res[jb & 1][i][e]), 0.f);
                    if (res_ok(jb, i)) *(f16x8*)(d.y + goff[i]) = o;
                }
            }
        }
        BN_STAMP(2)
        if (!has_next) break;
        tile = next;
        lds_barrier();                     // every wave has left the expand phase: the x buffers (r, patches) may be rewritten
        store_x(xa, xbuf0);
        lds_barrier();
    }
#ifdef GDT_BNECK_STAMP
    if (lane == 0 && d.stamps) { unsigned long long* o = d.stamps + ((size_t)bid * NWAVE + wave) * 4; o[0] = st_acc[0]; o[1] = st_acc[1]; o[2] = st_acc[2]; }
#endif
}

template <int C, int MID, int PH, int CIN = C, bool DS = false, bool RAGGED = false>
__global__ __launch_bounds__(NT) void conv_bneck_kernel(const BneckLaunch d, const int ntiles) {
    conv_bneck_body<C, MID, PH, CIN, DS, RAGGED>(d, ntiles, blockIdx.x, gridDim.x);
}
template <int C, int MID, int PH, int CIN = C, bool DS = false, bool RAGGED = false>
__global__ __launch_bounds__(NT) void conv_bneck_multi_kernel(const MultiBneck m) {
    const int l = gdt_multi_level(m.nlev, m.prefix, blockIdx.x);
    conv_bneck_body<C, MID, PH, CIN, DS, RAGGED>(m.lev[l], m.ntiles[l], blockIdx.x - m.prefix[l], m.prefix[l + 1] - m.prefix[l]);
}

template <int C, int MID, int PH, int CIN = C, bool DS = false, bool RAGGED = false>
int launch_multi(const BneckLaunch* dl, int L, hipStream_t stream) {
    using G = Geo<C, MID, PH, CIN, DS>;
    static GdtPerDevice per_dev;
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_bneck_multi_kernel<C, MID, PH, CIN, DS, RAGGED>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    MultiBneck m;
    m.nlev = L;
    int want[GDT_MAX_LEVELS];
    for (int l = 0; l < L; ++l) {
        m.lev[l] = dl[l];
        m.ntiles[l] = dl[l].N * ((dl[l].W + PW - 1) / PW) * ((dl[l].H + PH - 1) / PH);
        want[l] = (m.ntiles[l] + 7) / 8 * 8;
    }
    const int grid = gdt_multi_partition(m.prefix, want, L, cus);
    hipLaunchKernelGGL((conv_bneck_multi_kernel<C, MID, PH, CIN, DS, RAGGED>), dim3(grid), dim3(NT), G::LDS, stream, m);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

template <int C, int MID, int PH, int CIN = C, bool DS = false, bool RAGGED = false>
int launch(const BneckLaunch& d, hipStream_t stream) {
    using G = Geo<C, MID, PH, CIN, DS>;
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_bneck_kernel<C, MID, PH, CIN, DS, RAGGED>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    const int ntiles = d.N * ((d.W + PW - 1) / PW) * ((d.H + PH - 1) / PH);
    const int grid = min(cus, (ntiles + 7) / 8 * 8);
#ifdef GDT_BNECK_STAMP
    static unsigned long long* sb = nullptr;
    static int calls = 0;
    if (!sb) GDT_CHECK_HIP(hipMalloc((void**)&sb, (size_t)cus * NWAVE * 4 * 8));
    BneckLaunch ds = d; ds.stamps = sb;
    GDT_CHECK_HIP(hipMemsetAsync(sb, 0, (size_t)cus * NWAVE * 4 * 8, stream));
    hipLaunchKernelGGL((conv_bneck_kernel<C, MID, PH, CIN, DS, RAGGED>), dim3(grid), dim3(NT), G::LDS, stream, ds, ntiles);
    if (++calls % 50 < 3) {
        GDT_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)grid * NWAVE * 4);
        GDT_CHECK_HIP(hipMemcpy(h.data(), sb, h.size() * 8, hipMemcpyDeviceToHost));
        double a[3] = {0, 0, 0};
        for (size_t w = 0; w < (size_t)grid * NWAVE; ++w) for (int k = 0; k < 3; ++k) a[k] += (double)h[w * 4 + k];
        const double tiles_per_wg = (double)ntiles / grid, nw = (double)grid * NWAVE;
        fprintf(stderr, "[bneck stamp] C %d: per patch: reduce %.0f, 3x3 %.0f, expand %.0f cycles (%.1f patches per workgroup)\n", C, a[0] / nw / tiles_per_wg,
                a[1] / nw / tiles_per_wg, a[2] / nw / tiles_per_wg, tiles_per_wg);
    }
    return GDT_OK;
#endif
    hipLaunchKernelGGL((conv_bneck_kernel<C, MID, PH, CIN, DS, RAGGED>), dim3(grid), dim3(NT), G::LDS, stream, d, ntiles);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// Eligible: the three convs of an identity Bottleneck (checked by the planner in net.hip) with (C, MID) = (256, 64) or (512, 128), maps that
// the patches tile exactly, enough patches to fill the chip, offsets within 32 bits.
bool gdt_bneck_eligible(int cin, int C, int mid, int N, int H, int W) {
    const char* e = getenv("GDT_CONV_BNECK");          // 0: off (read at plan time, once per net and geometry: A/B inside one process)
    if (e && atoi(e) == 0) return false;
    int ph = 0;
    if (C == 256 && mid == 64 && (cin == 256 || cin == 64)) ph = 16;        // cin 64: the projection-shortcut form (layer1's first block)
    else if (C == 512 && mid == 128 && cin == 512) ph = 8;
    else return false;
    // patches that hang over the image (ragged form) may waste at most 15 % of the patch area
    if ((double)H * W / ((double)((H + ph - 1) / ph * ph) * ((W + PW - 1) / PW * PW)) < 0.85) return false;
    if ((long)N * H * W * C >= (1L << 31)) return false;
    static const int min_tiles = [] { const char* m = getenv("GDT_BNECK_MIN_TILES"); return m ? atoi(m) : 256; }();
    return (long)N * ((H + ph - 1) / ph) * ((W + PW - 1) / PW) >= min_tiles;
}

int gdt_launch_bneck(const f16* x, f16* y, const f16* wr, const f16* w3, const f16* we, const float* br, const float* b3, const float* be,
                     const f16* wd, const float* bd, int cin, int C, int mid, int N, int H, int W, hipStream_t stream) {
    BneckLaunch d{nullptr, x, y, wr, w3, we, br, b3, be, wd, bd, N, H, W};
    if (cin == 256 && C == 256 && mid == 64 && !wd) return (H % 16 || W % PW) ? launch<256, 64, 16, 256, false, true>(d, stream) : launch<256, 64, 16>(d, stream);
    if (cin == 512 && C == 512 && mid == 128 && !wd) return (H % 8 || W % PW) ? launch<512, 128, 8, 512, false, true>(d, stream) : launch<512, 128, 8>(d, stream);
    if (cin == 64 && C == 256 && mid == 64 && wd && bd) return (H % 16 || W % PW) ? launch<256, 64, 16, 64, true, true>(d, stream) : launch<256, 64, 16, 64, true>(d, stream);
    gdt_set_error("gdt_launch_bneck: unsupported shape");
    return GDT_ERR_INVALID;
}

// The same Bottleneck on L independent geometries (the levels of a pyramid) in ONE launch: every level its own x / y pointers and (N, H, W); the RAGGED
// instantiation when any level needs it (it is the general form: same results on maps the patches tile exactly)
int gdt_launch_bneck_levels(const f16* const* x, f16* const* y, const f16* wr, const f16* w3, const f16* we, const float* br, const float* b3, const float* be,
                            const f16* wd, const float* bd, int cin, int C, int mid, const int* N, const int* H, const int* W, int L, hipStream_t stream) {
    GDT_REQUIRE(L >= 1 && L <= GDT_MAX_LEVELS, "1..4 geometries per launch");
    if (L == 1) return gdt_launch_bneck(x[0], y[0], wr, w3, we, br, b3, be, wd, bd, cin, C, mid, N[0], H[0], W[0], stream);
    BneckLaunch dl[GDT_MAX_LEVELS];
    const int ph = (C == 512) ? 8 : 16;
    bool ragged = false;
    for (int l = 0; l < L; ++l) {
        dl[l] = BneckLaunch{nullptr, x[l], y[l], wr, w3, we, br, b3, be, wd, bd, N[l], H[l], W[l]};
        ragged = ragged || (H[l] % ph) || (W[l] % PW);
    }
    if (cin == 256 && C == 256 && mid == 64 && !wd) return ragged ? launch_multi<256, 64, 16, 256, false, true>(dl, L, stream) : launch_multi<256, 64, 16>(dl, L, stream);
    if (cin == 512 && C == 512 && mid == 128 && !wd) return ragged ? launch_multi<512, 128, 8, 512, false, true>(dl, L, stream) : launch_multi<512, 128, 8>(dl, L, stream);
    if (cin == 64 && C == 256 && mid == 64 && wd && bd) return ragged ? launch_multi<256, 64, 16, 64, true, true>(dl, L, stream) : launch_multi<256, 64, 16, 64, true>(dl, L, stream);
    gdt_set_error("gdt_launch_bneck_levels: unsupported shape");
    return GDT_ERR_INVALID;
}
