// ResNet Bottleneck tail in one launch (gfx950, MI355X):  y = ReLU(x + BN(W_e . ReLU(BN(W_3 (*) r))))  for MID = 256 (ResNet-101 layer3,
// torchvision Bottleneck as restated in oracle/gandtr_oracle.py:116-133; reference: external/cirtorch/networks/imageretrievalnet.py:189-190).
//
// Layer by layer the 3x3 conv (conv3x3_halo_rb.hip) writes the 256-channel tensor t and the expand conv (conv1x1_rb.hip) reads it back next to
// the 1024-channel residual: the first is MFMA-bound with HBM idle, the second HBM-bound with the matrix pipe at a quarter.  Here a workgroup
// keeps t for its 16 x 16 patch in LDS and runs the expand GEMM on it while the residual / output lines stream: one launch, t never exists in HBM,
// and the expand's HBM stream lies under MFMA work.
//
//   phase A (3x3): conv3x3_halo_rb's loop -- halo of r through registers into two LDS stages (64-channel chunks, XOR-swizzled 128-byte rows),
//                  W_3 fragments streamed L2 -> registers one step ahead, 8 waves as 2 (pixel halves) x 4 (64 channels), D = W . A^T.
//   hand-over:     t = ReLU(acc + b_3) rounded to fp16 (the same rounding the unfused path stores) goes to LDS as four planes of [256 pixels][64 k]
//                  (128 KB: they alias both halo stages), rows swizzled like the halo rows so that the fragment reads of phase B are conflict-free.
//   phase B (1x1): x_cout / 256 passes; per pass a wave owns 128 pixels x 64 output channels (the accumulator registers of phase A), K = 256 in 16
//                  k-steps, W_e fragments through the same four-slot register ring (the last slots of a pass already fetch the next pass's -- or the next
//                  tile's W_3 -- fragments); per-pass epilogue: 32 x 64 blocks transposed through a wave-private 4 KB patch (swizzled, no padding: mid
//                  planes + patches = 160 KiB to the byte), residual fetched and output stored as whole 128-byte lines; the residual loads of a row block
//                  are issued one block ahead (the first from inside the k-loop).
//                  The expand conv's bias enters as one more MFMA per accumulator (weight fragment { fp16(b), fp16(b - fp16(b)), 0 .. } against a pixel operand
//                  { 1, 1, 0 .. }): no bias registers in the epilogue, which holds the accumulators and two residual buffers.
//   The LDS has no room for the next tile's first halo chunk while t lives: it is staged after the last pass (two tiles per workgroup at the bench geometry).
//   Every lane-derived address of phase B is made from an opaque lane copy where it is used: hoisted out of the pass loop they were spilled (82 registers in
//   the first build), and a scratch reload is a vector-memory wait behind the HBM stream.
//
//   phase C (round 5, CHAIN form, PH = 8): the NEXT block's reduce conv  r' = ReLU(BN(W_r . y))  (1x1, x_cout -> 256: torchvision Bottleneck.conv1 of the following
//                  block) on the tile this workgroup has just written.  Its accumulators (128 pixels x 64 channels per wave = 128 registers) cannot live beside
//                  phase B's (the K of this GEMM is produced pass by pass, so they would have to stay live through all of phase B: 256 accumulator registers + 100 of
//                  operands against the 256 a wave has at two waves per SIMD; one wave per SIMD would hold them but gives up the second workgroup that overlaps the
//                  phases), and the LDS has no room for the tile of y (256 KB).  So the tile's y lines are read back -- written microseconds ago by this very
//                  workgroup, they come from L2 / the Infinity Cache, not from HBM -- in 64-channel chunks through registers into two of the (dead) t planes, the
//                  K-loop is conv1x1_rb's (W_r fragments through the four-slot ring, activations one chunk ahead), r' leaves through the wave-private patches as
//                  whole 128-byte lines.  The separate reduce launch (0.08 ms per block, 268 MB of y read back from HBM) is gone.
//                  Measured (round 5): the chained launch takes 0.305 ms against 0.227 + 0.083 ms for the two launches it replaces -- a tie in time (2.40 k
//                  descriptors/s either way), 5.6 GB less HBM traffic per forward and 21 launches fewer.  Stamps per tile and wave: phase A 75 k cycles (36.9 k of
//                  MFMA time: two workgroups share each SIMD), hand-over 6 k, phase B k-loops 33 k, phase B epilogues 44 k, phase C 75 k (16.4 k of MFMA time), next
//                  tile's first chunk 12 k.  Ablations of phase C (timing only): without its y loads 59 k, without its weight re-loads 60 k, without its per-chunk
//                  barriers 72 k, with none of the three 46 k (= prologue + epilogue + a k-loop like phase B's).  The launch as a whole keeps the matrix pipes 58 %
//                  busy (70 k of MFMA time per tile in the 120 k cycles a CU spends per tile); what is missing is a third wave per SIMD to issue while two wait,
//                  and 256-register waves leave room for two.
//
// Measured (GeM-ResNet-101, 32 x 1024^2, layer3: 131 072 pixels per block): layer by layer 0.128 + 0.143 ms; one 512-thread workgroup per CU on 16 x 16
// patches 0.240 ms; two 256-thread workgroups per CU on 8 x 16 patches, the second started 30 us late, 0.222 ms (stamps: phase A 99 k cycles per 256 pixels
// without a byte of HBM traffic, phase B 87 k cycles for 1 MB per tile = the CU's share of ~5 TB/s: with every CU in the same phase at the same time the two
// bounds ADD; two workgroups per CU half a tile apart overlap them -- and the clock drops from 1.67 to 1.4 GHz: the forward runs against the board's power
// limit, like the generator's).
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "gdt_common.h"

#ifndef GDT_XEXP_CABL
#define GDT_XEXP_CABL 0        // timing-only ablations of phase C (results are wrong by design): 1 no y loads / LDS writes, 2 no per-chunk barrier, 4 no weight re-loads
#endif
#ifndef GDT_XEXP_CDEPTH
#define GDT_XEXP_CDEPTH 1      // phase C: register sets of y chunks in flight (1, 3 or 5).  Measured: 3 sets change nothing (phase C 75.4 k cycles per tile either way,
                               // 7-11 spilled registers) -- vector loads retire in order, so the wait for a weight fragment issued after a y load also waits for that y
                               // load, however early the NEXT one was requested: the latency a y line may take is the depth of the weight ring (four k-steps)
#endif

namespace {

constexpr int ROWB = 128;                                   // bytes per LDS row (64 halves of K)
constexpr int HW_ = 18;
constexpr int PATCH_BYTES = 32 * 64 * 2;
constexpr int NCH = 4;                                      // 64-channel chunks of r (MID = 256)
constexpr int TM = 4, TN = 2;

// PH = 16: one 512-thread workgroup per CU on 16 x 16 patches (8 waves: 2 pixel halves x 4 channel quarters).
// PH = 8:  TWO 256-thread workgroups per CU on 8 x 16 patches (4 waves: 4 channel quarters), 80 KB of LDS each.  Phase A moves no HBM bytes and phase B is
//          HBM-bound; with one workgroup per CU every CU of the chip is in the same phase at the same time (two tiles per workgroup, equal work: lockstep), so
//          the matrix pipes idle through B and HBM through A.  Two workgroups per CU that start half a tile apart (GDT_XEXP_STAGGER_US) keep one in each phase.
template <int PH> struct Geo {
    static constexpr int WAVES = PH / 2, NT = WAVES * 64, RPR = NT / 8;
    static constexpr int HROWS = (PH + 2) * HW_, HROWS_PAD = (HROWS + 7) / 8 * 8;
    static constexpr int A_BYTES = HROWS_PAD * ROWB;            // one halo stage (41 984 / 23 552 B)
    static constexpr int NR = (HROWS_PAD + RPR - 1) / RPR;      // loader rounds per chunk (6)
    static constexpr int PLANE = PH * 16 * ROWB;                // one 64-k plane of t (32 / 16 KB)
    static constexpr int PATCH_OFF = 4 * PLANE;
    static constexpr int LDS_BYTES = PATCH_OFF + WAVES * PATCH_BYTES;      // 163 840 / 81 920
    static_assert(2 * A_BYTES <= PATCH_OFF && LDS_BYTES * (PH == 16 ? 1 : 2) <= 160 * 1024 && NR <= 8, "LDS plan / staging schedule");
};

struct TileAt { int n, y0, x0; bool valid; };

struct Pend4 { f16x8 v[4]; };

template <int PH, bool CHAIN = false>
__global__ __launch_bounds__(Geo<PH>::NT, PH == 16 ? 1 : 2) void conv3x3_expand_rb_kernel(const ConvLaunch d, const int ntiles) {
    using G = Geo<PH>;
    static_assert(!CHAIN || PH == 8, "the chained form exists for the two-workgroups-per-CU layout only");
    constexpr int NT = G::NT, RPR = G::RPR, HROWS = G::HROWS, HROWS_PAD = G::HROWS_PAD, A_BYTES = G::A_BYTES, NR = G::NR, PLANE = G::PLANE, PATCH_OFF = G::PATCH_OFF;
    (void)NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                   // (PH = 8: wm = 0)

    // the second workgroup of a CU starts late: an XCD's workgroups are dispatched CU by CU, so the second half of an XCD lane's indices are the second round
    // (stagger_us < 0, dev: every other index instead)
    const int xj = blockIdx.x >> 3, per_xcd = gridDim.x >> 4;
    if (PH == 8 && d.stagger_us != 0 && (d.stagger_us > 0 ? xj >= per_xcd : (xj & 1))) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), wait = (unsigned long long)(d.stagger_us > 0 ? d.stagger_us : -d.stagger_us) * 1700;     // s_memtime counts shader cycles (~1.7 GHz in this kernel)
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
    const int tiles_x = (d.W + 15) >> 4, tiles_y = (d.H + PH - 1) / PH, tpi = tiles_x * tiles_y;
    auto tile_at = [&](int vb) -> TileAt {
        TileAt t;
        int tile_m = 0, tile_n = 0;
        t.valid = vb < 8 * ((ntiles + 7) / 8) && gdt_tile_of_block(vb, ntiles, 1, tile_m, tile_n);
        if (!t.valid) tile_m = 0;
        t.n = tile_m / tpi;
        const int tr = tile_m - t.n * tpi;
        t.y0 = (tr / tiles_x) * PH; t.x0 = (tr % tiles_x) << 4;
        return t;
    };
    int vb = blockIdx.x;
    TileAt cur = tile_at(vb);
    if (!cur.valid) return;                   // (G is a multiple of 8: a workgroup's tiles stay on its XCD lane and validity is monotone along b, b + G, ...)

    // ---- halo loader of phase A (conv3x3_halo_rb.hip: through registers, branch-free, zero padding)
    const int lrow = tid >> 3;
    // a piece = 16 bytes of one halo row; the loader and the LDS write both derive (row, validity) from (tile, round): only the data waits in registers
    auto piece_at = [&](const TileAt& ta, int r, int& h, int& hx, int& ry, int& rx) -> bool {
        int lr = lrow;
        asm volatile("" : "+v"(lr));                                   // (keeps the address arithmetic inside the loop: hoisted, it spills)
        h = min(r * RPR + lr, HROWS_PAD - 1);
        const int hy = (h * 3641) >> 16;
        hx = h - hy * HW_;
        const int iy = ta.y0 - 1 + hy, ix = ta.x0 - 1 + hx;
        ry = min(max(iy, 0), d.H - 1); rx = min(max(ix, 0), d.W - 1);
        return (h < HROWS) & ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
    };
    auto load_piece = [&](const TileAt& ta, int chunk, int r) -> f16x8 {
        int h, hx, ry, rx;
        piece_at(ta, r, h, hx, ry, rx);
        const int q = (lane & 7) ^ ((hx >> 1) & 7);
        const unsigned goff = ((unsigned)((ta.n * d.H + ry) * d.W + rx) << 8) + (chunk * 8 + q) * 8;
        return *(const f16x8*)(d.in + goff);
    };
    auto store_piece = [&](const TileAt& ta, int stage_off, int r, const f16x8& raw) {
        int row, hx, ry, rx;
        const bool ok = piece_at(ta, r, row, hx, ry, rx);
        f16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (f16)0.f;
        int l7 = lane & 7;
        asm volatile("" : "+v"(l7));
        *(f16x8*)(smem + stage_off + row * ROWB + (l7 << 4)) = ok ? raw : z;
    };

    // ---- weights: fragment order [cout / 32][K / 16][lane][8]; ONE four-slot ring for both phases
    constexpr int NKS3 = 9 * 256 / 16, NKSE = 256 / 16;
    // (every weight load = uniform base + an OPAQUE copy of the 32-bit lane offset: hoisted out of the persistent loop, the 64-bit per-lane pointers of the
    // three weight arrays get spilled, and each scratch reload is a vector-memory wait behind the HBM stream)
    auto lane_off_now = [&]() -> unsigned { unsigned v = lane * 8; asm volatile("" : "+v"(v)); return v; };
    const f16* w3 = d.w_frag + (long)(wn * 2) * NKS3 * 512;                        // this wave's two 32-channel blocks of W_3
    f16x8 b[4][TN];
    auto load_b3 = [&](int kk, int kstep) {
        const unsigned lane_off = lane_off_now();
#pragma unroll
        for (int j = 0; j < TN; ++j) b[kk][j] = *(const f16x8*)(w3 + ((long)j * NKS3 + kstep) * 512 + lane_off);
    };
    auto load_be = [&](int kk, int pass, int ks) {
        const f16* we = d.x_w_frag + (long)((pass * 8 + wn * 2) * NKSE + ks) * 512;
        const unsigned lane_off = lane_off_now();
#pragma unroll
        for (int j = 0; j < TN; ++j) b[kk][j] = *(const f16x8*)(we + (long)j * NKSE * 512 + lane_off);
    };

    constexpr int NKSR = 1024 / 16;                                                // (CHAIN: K of the next block's reduce conv = x_cout = 1024)
    // ---- fragment addresses
    const int fr = lane & 31, fh = lane >> 5;
    int vt[3];                                                                      // phase A: per tap column (the swizzle depends on px + tx)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) vt[tx] = ((wm * 8 + (fr >> 4)) * HW_ + (fr & 15)) * ROWB + ((fh ^ ((((fr & 15) + tx) >> 1) & 7)) << 4);
    auto a_frag = [&](int stage_off, int i, int ty, int tx, int kk) -> f16x8 {
        return *(const f16x8*)(smem + ((vt[tx] + stage_off) ^ (kk << 5)) + (i * 2 * HW_ + ty * HW_ + tx) * ROWB);
    };

    const int npass = d.x_cout >> 8;
    f16x8 afr[2][TM];
    // ---- prologue: W_3 of step 0, halo chunk 0 of the first tile
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) load_b3(kk, kk);
#pragma unroll
    for (int r = 0; r < NR; ++r) store_piece(cur, 0, r, load_piece(cur, 0, r));
    __syncthreads();

#ifdef GDT_XEXP_STAMP
    unsigned long long st_a = 0, st_ho = 0, st_bk = 0, st_be = 0, st_nx = 0, st_c = 0, st_t = __builtin_amdgcn_s_memtime(), st_n = 0;
    const unsigned long long st_begin = st_t;
#define GDT_STAMP(acc_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - st_t; st_t = now_; }
#else
#define GDT_STAMP(acc_)
#endif
    for (;;) {
        TileAt nxt = tile_at(vb + gridDim.x);
#pragma unroll
        for (int i = 0; i < TM; ++i) afr[0][i] = a_frag(0, i, 0, 0, 0);
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // ------------------------------------------------------------ phase A: t = W_3 (*) r on the patch
        f16x8 pend;                                  // (overwritten before its first use)
#pragma unroll
        for (int e = 0; e < 8; ++e) pend[e] = (f16)0.f;
        int so = 0;
        for (int c = 0; c < NCH; ++c) {
            const bool last = c + 1 == NCH;
            const int sc = last ? 0 : c + 1;         // (the last chunk stages chunk 0 once more into the dead stage: idempotent, never read, keeps the loop straight-line)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = t / 3, tx = t - ty * 3;
                const int nty = (t + 1) / 3, ntx = (t + 1) - nty * 3;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int cu = kk & 1, nx = cu ^ 1;
                    if (kk < 3) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) afr[nx][i] = a_frag(so, i, ty, tx, kk + 1);
                    } else if (t < 8) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) afr[nx][i] = a_frag(so, i, nty, ntx, 0);
                    }
                    if (kk == 2) {                              // halo of the next chunk: one piece per step, written a step after its load
                        if (t >= 1 && t - 1 < NR) store_piece(cur, A_BYTES - so, t - 1, pend);
                        if (t < NR) pend = load_piece(cur, sc, t);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[kk][j], afr[cu][i], acc[i][j], 0, 0, 0);      // D[cout][pixel]
                    // the slot's next use: the same kk of the next step -- or, after the last step, the first k-steps of W_e
                    if (t < 8) load_b3(kk, (t + 1) * 16 + c * 4 + kk);
                    else {
                        const f16* nb = last ? d.x_w_frag + (long)(wn * 2 * NKSE + kk) * 512 : w3 + (long)((c + 1) * 4 + kk) * 512;        // (uniform select)
                        const long js = last ? (long)NKSE * 512 : (long)NKS3 * 512;
                        const unsigned lane_off = lane_off_now();
#pragma unroll
                        for (int j = 0; j < TN; ++j) b[kk][j] = *(const f16x8*)(nb + j * js + lane_off);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            so = A_BYTES - so;
            if (!last) {
#pragma unroll
                for (int i = 0; i < TM; ++i) afr[0][i] = a_frag(so, i, 0, 0, 0);
            }
        }

        GDT_STAMP(st_a)
        // ------------------------------------------------------------ hand-over (all waves are done with the halo stages: the barrier above)
        int fr_e = fr, fh_e = fh, lane_e = lane;                   // opaque copies: the epilogue's addresses stay out of the persistent loop's invariants
        asm volatile("" : "+v"(fr_e), "+v"(fh_e), "+v"(lane_e));
        {
            char* mrow = smem + wn * PLANE + (wm * 128 + fr_e) * ROWB + fh_e * 8;
            const int sw = (fr_e >> 1) & 7;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *(const float4*)(d.bias + wn * 64 + j * 32 + 8 * g + 4 * fh_e);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const f32x16& a = acc[i][j];
                        f16x4 h;
                        h[0] = (f16)fmaxf(a[4 * g] + bv.x, 0.f); h[1] = (f16)fmaxf(a[4 * g + 1] + bv.y, 0.f);
                        h[2] = (f16)fmaxf(a[4 * g + 2] + bv.z, 0.f); h[3] = (f16)fmaxf(a[4 * g + 3] + bv.w, 0.f);
                        *(f16x4*)(mrow + i * (32 * ROWB) + (((j * 4 + g) ^ sw) << 4)) = h;
                    }
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        GDT_STAMP(st_ho)
        // ------------------------------------------------------------ phase B: y = ReLU(x + W_e . t + b_e), 256 output channels per pass
        f16* patch = (f16*)(smem + PATCH_OFF + wave * PATCH_BYTES);
        const int vm = (wm * 128 + fr_e) * ROWB + ((fh_e ^ ((fr_e >> 1) & 7)) << 4);        // t fragments: row = pixel of the patch (made here: phase A does not carry it)
        auto m_frag = [&](int i, int ks) -> f16x8 {
            return *(const f16x8*)(smem + (vm ^ ((ks & 3) << 5)) + (ks >> 2) * PLANE + i * (32 * ROWB));
        };
        auto opq = [](int v) -> int { asm volatile("" : "+v"(v)); return v; };       // a fresh opaque copy: what is derived from it cannot be hoisted out of the pass loop (and spilled)
        struct Res { f16x8 v[4]; };
        auto res_offs = [&](int i, int q, int pass, bool& ok) -> unsigned {
            const int le = opq(lane_e);
            const int px = (le >> 3) + 8 * q;
            const int y = cur.y0 + wm * 8 + 2 * i + (px >> 4), x = cur.x0 + (px & 15);         // (wm * 8: second pixel half of a 16-row patch)
            ok = (y < d.H) & (x < d.W);
            const int ch = le & 7;
            return ok ? (unsigned)(((cur.n * d.H + y) * d.W + x) * d.x_cout + pass * 256 + wn * 64 + ch * 8) : 0u;
        };
        auto load_res = [&](int i, int pass) -> Res {
            Res r;
#pragma unroll
            for (int q = 0; q < 4; ++q) { bool ok; r.v[q] = *(const f16x8*)(d.res + res_offs(i, q, pass, ok)); }
            return r;
        };
        f16x8 bb[TN];                                // bias fragments of the pass (fetched during the previous pass's k-loop)
#pragma unroll
        for (int j = 0; j < TN; ++j) bb[j] = *(const f16x8*)(d.x_bias + (long)(wn * 2 + j) * 512 + lane_off_now());
        for (int pass = 0; pass < npass; ++pass) {
            const bool lastp = pass + 1 == npass;
#pragma unroll
            for (int i = 0; i < TM; ++i) afr[0][i] = m_frag(i, 0);
            // the bias step initialises the accumulators: acc = b_e (x) 1   (pixel operand: k 0, 1 = 1 in the lanes of the low k half)
            f16x8 ones;
            {
                const int fhe = opq(fh_e);
#pragma unroll
                for (int e = 0; e < 8; ++e) ones[e] = (f16)((e < 2 && fhe == 0) ? 1.f : 0.f);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 z;
#pragma unroll
                    for (int e = 0; e < 16; ++e) z[e] = 0.f;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bb[j], ones, z, 0, 0, 0);
                }
            {   // next pass's bias fragments (the last pass fetches the first pass's again: unconditional, never used)
                const int np = lastp ? 0 : pass + 1;
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[j] = *(const f16x8*)(d.x_bias + (long)(np * 8 + wn * 2 + j) * 512 + lane_off_now());
            }
            Res rv0;
#pragma unroll
            for (int ks = 0; ks < NKSE; ++ks) {
                const int kk = ks & 3, cu = ks & 1, nx = cu ^ 1;
                if (ks + 1 < NKSE) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) afr[nx][i] = m_frag(i, ks + 1);
                }
                if (ks == 8) rv0 = load_res(0, pass);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[kk][j], afr[cu][i], acc[i][j], 0, 0, 0);
                if (ks + 4 < NKSE) load_be(kk, pass, ks + 4);
                else {
                    // next pass's first k-steps, or W_3's first step for the next tile -- CHAIN: W_r's first k-steps (uniform select: the loop stays straight-line)
                    const f16* nb = lastp ? (CHAIN ? d.r_w_frag + (long)(wn * 2 * NKSR + kk) * 512 : w3 + (long)kk * 512)
                                          : d.x_w_frag + (long)(((pass + 1) * 8 + wn * 2) * NKSE + kk) * 512;
                    const long js = lastp ? (CHAIN ? (long)NKSR * 512 : (long)NKS3 * 512) : (long)NKSE * 512;
                    const unsigned lane_off = lane_off_now();
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[kk][j] = *(const f16x8*)(nb + j * js + lane_off);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            GDT_STAMP(st_bk)
            // ---- epilogue of the pass: per 32-pixel row block transpose through the wave's patch, + residual, ReLU, whole-line stores
            Res rv1;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (i + 1 < TM) { if (i & 1) rv0 = load_res(i + 1, pass); else rv1 = load_res(i + 1, pass); }
                const Res& rv = (i & 1) ? rv1 : rv0;
                // patch addresses: ONE per-lane base, the channel chunk XOR-ed in at the use (the row swizzle sits in bits 4-6, which the base leaves clear)
                const int fre = opq(fr_e), fhe = opq(fh_e);
                const int wbase = fre * 128 + (((fre >> 1) & 7) << 4) + fhe * 8;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x16& a = acc[i][j];
                        f16x4 h;
                        h[0] = (f16)a[4 * g]; h[1] = (f16)a[4 * g + 1]; h[2] = (f16)a[4 * g + 2]; h[3] = (f16)a[4 * g + 3];
                        *(f16x4*)((char*)patch + (wbase ^ ((j * 4 + g) << 4))) = h;
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int le = opq(lane_e);
                    const int px = (le >> 3) + 8 * q, ch = le & 7;
                    bool ok;
                    const unsigned off = res_offs(i, q, pass, ok);
                    f16x8 v = *(const f16x8*)(patch + px * 64 + ((ch ^ ((px >> 1) & 7)) << 3));
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)fmaxf((float)v[e] + (float)rv.v[q][e], 0.f);
                    if (ok) *(f16x8*)(d.out + off) = v;
                }
            }
#ifdef GDT_XEXP_STAMP
#if GDT_XEXP_STAMP > 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (diagnostic only: the epilogue's stores are charged to the epilogue)
#endif
            GDT_STAMP(st_be)
#endif
        }
        if (CHAIN) {
            // ------------------------------------------------------------ phase C: r' = ReLU(W_r . y + b_r) on this tile (the next block's reduce conv)
            // the y lines of this tile were stored by all four waves: complete (vmcnt) and visible workgroup-wide before anyone reads them back; the same barrier
            // retires the last reads of the t planes, two of which take the y chunks
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            auto c_load = [&](int chunk) -> Pend4 {
                Pend4 pp;
                int lr = lrow, l7 = lane & 7;
                asm volatile("" : "+v"(lr), "+v"(l7));
                const int q = l7 ^ ((lr >> 1) & 7);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int px = r * 32 + lr;                                       // row of the plane = pixel of the 8 x 16 patch
                    const int yy = min(cur.y0 + (px >> 4), d.H - 1), xx = min(cur.x0 + (px & 15), d.W - 1);      // (pixels past the image: a valid line, its result is not stored)
                    pp.v[r] = *(const f16x8*)(d.out + (unsigned)((cur.n * d.H + yy) * d.W + xx) * (unsigned)d.x_cout + (unsigned)(chunk * 64 + q * 8));
                }
                return pp;
            };
            auto c_store = [&](const Pend4& pp, int plane) {
                int lr = lrow, l7 = lane & 7;
                asm volatile("" : "+v"(lr), "+v"(l7));
#pragma unroll
                for (int r = 0; r < 4; ++r) *(f16x8*)(smem + plane * PLANE + (r * 32 + lr) * ROWB + (l7 << 4)) = pp.v[r];
            };
            const int vc = opq(fr_e) * ROWB + ((opq(fh_e) ^ ((opq(fr_e) >> 1) & 7)) << 4);
            auto c_frag = [&](int plane, int i, int kk) -> f16x8 {
                return *(const f16x8*)(smem + (vc ^ (kk << 5)) + plane * PLANE + i * (32 * ROWB));
            };
            // CD register sets take turns (static indices): chunk c + 1 + CD is requested at the end of chunk c and written to LDS in chunk c + CD
            constexpr int CD = GDT_XEXP_CDEPTH;
            Pend4 P[CD];
            { const Pend4 p0 = c_load(0); c_store(p0, 0); }
#pragma unroll
            for (int k = 0; k < CD; ++k) P[k] = c_load(k + 1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i) afr[0][i] = c_frag(0, i, 0);
            constexpr int NCC = 1024 / 64;                                               // chunks of y
            auto chunk_step = [&](Pend4& Pk, const int c, auto last_tag) {
                constexpr bool lastc = decltype(last_tag)::value;
                const int pl = c & 1;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int cu = kk & 1, nx = cu ^ 1;
                    if (kk < 3) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) afr[nx][i] = c_frag(pl, i, kk + 1);
                    }
                    if (kk == 1 && !lastc && !(GDT_XEXP_CABL & 1)) c_store(Pk, pl ^ 1);      // chunk c + 1
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[kk][j], afr[cu][i], acc[i][j], 0, 0, 0);
                    if (lastc || !(GDT_XEXP_CABL & 4)) {   // the slot's next use: the same kk of the next chunk -- or W_3's first step for the next tile
                        const f16* nb = lastc ? w3 + (long)kk * 512 : d.r_w_frag + (long)(wn * 2 * NKSR + (c + 1) * 4 + kk) * 512;
                        const long js = lastc ? (long)NKS3 * 512 : (long)NKSR * 512;
                        const unsigned lane_off = lane_off_now();
#pragma unroll
                        for (int j = 0; j < TN; ++j) b[kk][j] = *(const f16x8*)(nb + j * js + lane_off);
                    }
                    if (kk == 3 && !lastc && !(GDT_XEXP_CABL & 1)) Pk = c_load(min(c + 1 + CD, NCC - 1));      // behind the weights (in-order vmcnt); past the end: a line nobody uses
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lastc || !(GDT_XEXP_CABL & 2)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (!lastc) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) afr[0][i] = c_frag(pl ^ 1, i, 0);
                }
            };
            static_assert((NCC - 1) % CD == 0 && CD >= 1 && CD <= 3, "the register sets take turns over whole groups of chunks + the last one");
            for (int c0 = 0; c0 + CD < NCC; c0 += CD) {
#pragma unroll
                for (int k = 0; k < CD; ++k) chunk_step(P[k], c0 + k, std::false_type());
            }
            chunk_step(P[0], NCC - 1, std::true_type());
            // epilogue: + b_r, ReLU, fp16, per 32-pixel row block through the wave's patch, whole 128-byte lines of r'
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int fre = opq(fr_e), fhe = opq(fh_e);
                const int wbase = fre * 128 + (((fre >> 1) & 7) << 4) + fhe * 8;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bv = *(const float4*)(d.r_bias + wn * 64 + j * 32 + 8 * g + 4 * fhe);
                        const f32x16& a = acc[i][j];
                        f16x4 h;
                        h[0] = (f16)fmaxf(a[4 * g] + bv.x, 0.f); h[1] = (f16)fmaxf(a[4 * g + 1] + bv.y, 0.f);
                        h[2] = (f16)fmaxf(a[4 * g + 2] + bv.z, 0.f); h[3] = (f16)fmaxf(a[4 * g + 3] + bv.w, 0.f);
                        *(f16x4*)((char*)patch + (wbase ^ ((j * 4 + g) << 4))) = h;
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int le = opq(lane_e);
                    const int px = (le >> 3) + 8 * q, ch = le & 7;
                    const int y = cur.y0 + 2 * i + (px >> 4), x = cur.x0 + (px & 15);
                    const f16x8 v = *(const f16x8*)(patch + px * 64 + ((ch ^ ((px >> 1) & 7)) << 3));
                    if ((y < d.H) & (x < d.W)) *(f16x8*)(d.r_out + (unsigned)((cur.n * d.H + y) * d.W + x) * 256u + (unsigned)(wn * 64 + ch * 8)) = v;
                }
            }
            GDT_STAMP(st_c)
        }
#ifdef GDT_XEXP_STAMP
        ++st_n;
#endif
        if (!nxt.valid) break;
        // t is dead once every wave has left the last k-loop; the LDS had no room for the next tile's first halo chunk until now
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < NR; ++r) store_piece(nxt, 0, r, load_piece(nxt, 0, r));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cur = nxt; vb += gridDim.x;
        GDT_STAMP(st_nx)
    }
#ifdef GDT_XEXP_STAMP
    if (lane == 0 && d.stamp_out) {
        unsigned long long* o = d.stamp_out + ((long)blockIdx.x * G::WAVES + wave) * 8;
        o[0] = st_a; o[1] = st_ho; o[2] = st_bk; o[3] = st_be; o[4] = st_nx; o[5] = st_n; o[6] = __builtin_amdgcn_s_memtime() - st_begin; o[7] = st_c;
    }
#endif
}

}  // namespace

// Eligible: 3x3 / stride 1 / zero pad 1, 256 -> 256 channels with fragment-ordered weights and a bias (BatchNorm folded), ReLU; expand 1x1 to a multiple of 256
// channels with bias, residual and ReLU; enough patches for one per CU; at most 15 % of the patch area hanging over the image.
static int xexp_patch_height() {
    static const int ph = [] { const char* e = getenv("GDT_XEXP_PH"); return e ? atoi(e) : 8; }();       // 16: one 512-thread workgroup per CU on 16 x 16 patches
    return ph == 16 ? 16 : 8;
}

bool gdt_conv3x3_expand_eligible(const ConvLaunch& d) {
    if (!d.w_frag || !d.x_w_frag || !d.bias || !d.x_bias || !d.res || !d.out || d.out_f32) return false;
    if (d.Cin != 256 || d.Cout != 256 || d.CoutPad != 256 || d.x_cout < 256 || d.x_cout % 256 != 0) return false;
    if (d.ntaps != 9 || d.sy != 1 || d.sx != 1 || d.pad_reflect || d.in_norm || d.in_res || d.in_out || d.stats || d.pool2 || d.phase_cout || !d.relu) return false;
    if ((long)d.N * d.H * d.W * d.x_cout >= (1L << 32)) return false;                                  // 32-bit element offsets
    // (both figures for the patch height that WILL be launched: PH x 16 patches, 16 / PH workgroups per CU -- GDT_XEXP_MIN_TILES counts 16 x 16 patches' worth of work)
    const int ph = xexp_patch_height();
    const long tiles = (long)d.N * ((d.W + 15) / 16) * ((d.H + ph - 1) / ph);
    const double useful = (double)d.H * d.W / ((double)((d.H + ph - 1) / ph * ph) * ((d.W + 15) / 16 * 16));
    static const int min_tiles = [] { const char* e = getenv("GDT_XEXP_MIN_TILES"); return e ? atoi(e) : 256; }();
    if (d.group_factor > 1.f) {
        // one of several geometries in flight together (pyramid levels on side streams): what has to fill the chip is the group.  Measured, GeM-ResNet-101 hub
        // scales, levels concurrent: 8 x 1024^2 (128 + 72 + 32 patches of 16 x 16) 7.40 -> 6.72 ms with the fused launches, 4 x 1024^2 (116 patches) 3.88 -> 4.40,
        // 2 x 3.16 -> 3.72, 1 x 2.85 -> 3.15: fused from ~200 patches in the group, and never below 16 of its own
        static const int group_min = [] { const char* e = getenv("GDT_XEXP_GROUP_MIN_TILES"); return e ? atoi(e) : 192; }();
        return (double)tiles * ph * d.group_factor >= (double)group_min * 16 && tiles * ph >= 16 * 16 && useful >= 0.85;
    }
    return tiles * ph >= (long)min_tiles * 16 && useful >= 0.85;
}

// CHAIN form (phase C): the next block's reduce conv -- a 1x1 conv x_cout = 1024 -> 256 with bias and ReLU on the tensor this launch writes
bool gdt_conv3x3_expand_chain_eligible(const ConvLaunch& d) {
    const char* e = getenv("GDT_XEXP_CHAIN");                  // 0: the reduce conv stays its own launch (read when a net plans a geometry: A/B inside one process)
    return !(e && atoi(e) == 0) && xexp_patch_height() == 8 && d.x_cout == 1024 && gdt_conv3x3_expand_eligible(d);
}

template <int PH, bool CHAIN = false>
static int launch_xexp(const ConvLaunch& d_in, hipStream_t stream) {
    using G = Geo<PH>;
    ConvLaunch d = d_in;
    const int tiles = d.N * ((d.W + 15) / 16) * ((d.H + PH - 1) / PH);
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int cus = 0;
    {
        const int rc = gdt_per_device(per_dev, cus, [](int, int ncu, int& v) {
            v = ncu / 8 * 8;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_expand_rb_kernel<PH, CHAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    const int vblocks = gdt_grid_for_tiles(tiles, 1), slots = cus * (PH == 16 ? 1 : 2);
    int grid = vblocks < slots ? vblocks : slots;
    static const int cu_limit = [] { const char* e = getenv("GDT_CU_LIMIT"); return e ? atoi(e) : 0; }();      // dev: persistent grid on part of the chip (concurrent-stream experiments)
    if (cu_limit > 0 && grid > cu_limit) grid = cu_limit;
    static const int stagger = [] { const char* e = getenv("GDT_XEXP_STAGGER_US"); return e ? atoi(e) : 30; }();
    // a workgroup with a single patch has no steady state to de-phase, the delay is then a plain loss: GeM-ResNet-101 hub pyramid, 8 x 1024^2 per call (256 + 144 + 64
    // patches) 6.72 -> 6.36 ms without it, 16 x 12.87 -> 12.69
    d.stagger_us = tiles > slots ? stagger : 0;
#ifdef GDT_XEXP_STAMP
    constexpr int W = G::WAVES;
    static unsigned long long* stamp_buf = nullptr;
    static int stamp_calls = 0;
    if (!stamp_buf) GDT_CHECK_HIP(hipMalloc((void**)&stamp_buf, (size_t)cus * 2 * W * 8 * sizeof(unsigned long long)));
    GDT_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, (size_t)cus * 2 * W * 8 * sizeof(unsigned long long), stream));
    d.stamp_out = stamp_buf;
    hipLaunchKernelGGL((conv3x3_expand_rb_kernel<PH, CHAIN>), dim3(grid), dim3(G::NT), G::LDS_BYTES, stream, d, tiles);
    if (++stamp_calls % 100 < 4) {
        GDT_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)grid * W * 8);
        GDT_CHECK_HIP(hipMemcpy(h.data(), stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double s[7] = {0, 0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)grid * W; ++w) for (int k = 0; k < 7; ++k) s[k] += (double)h[w * 8 + k];
        const double nw = (double)grid * W, nt = s[5] / nw;
        double sc = 0;
        for (size_t w = 0; w < (size_t)grid * W; ++w) sc += (double)h[w * 8 + 7];
        fprintf(stderr, "[xexp stamp] PH %d chain %d tiles/wave %.1f; per tile: phase A %.0f, hand-over %.0f, phase B k-loops %.0f, phase B epilogues %.0f, phase C %.0f, next-tile staging %.0f (per wave) cycles; total per wave %.0f\n",
                PH, (int)CHAIN, nt, s[0] / nw / nt, s[1] / nw / nt, s[2] / nw / nt, s[3] / nw / nt, sc / nw / nt, s[4] / nw, s[6] / nw);
    }
#else
    hipLaunchKernelGGL((conv3x3_expand_rb_kernel<PH, CHAIN>), dim3(grid), dim3(G::NT), G::LDS_BYTES, stream, d, tiles);
#endif
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

int gdt_launch_conv3x3_expand(const ConvLaunch& d, hipStream_t stream) {
    if (d.r_w_frag) {
        GDT_REQUIRE(d.r_bias && d.r_out && gdt_conv3x3_expand_chain_eligible(d), "chained reduce conv: 1024 -> 256 with bias, 8 x 16 patches");
        return launch_xexp<8, true>(d, stream);
    }
    return xexp_patch_height() == 16 ? launch_xexp<16>(d, stream) : launch_xexp<8>(d, stream);
}
