// Streaming 1x1 convolution for the ResNet-101 Bottlenecks (gfx950, MI355X).
//
// The Bottleneck 1x1 convs (torchvision resnet.py Bottleneck.conv1 / conv3 as sliced at imageretrievalnet.py:185-190; BN folded
// by the packer) are HBM-bound GEMMs: per 256-pixel tile the 256 -> 1024 expand moves 288 KB (12 us at a CU's share of the
// achievable 6 TB/s) for 3.4 us of MFMA work.  conv_igemm_rb.hip runs them as ONE 8-wave workgroup per CU at the register
// limit (256 VGPRs): nothing hides the latency of the residual reads and of the store acknowledgements its next weight loads
// queue behind (in-order vmcnt), and its A loads get a single K-step to land because the weight loads of the next step are
// issued after them.  Measured: 3.5 TB/s on the layer3 convs.  Here the same data path is cut for thread-level parallelism:
//   * 128 x 128 tile, four waves of 64 x 64 (64 accumulator registers): two to three workgroups per CU, each at a different
//     point of its tile -- one's epilogue (residual reads, 128-byte line stores) runs under the others' K-loops;
//   * activations two to four K-steps ahead in register sets that take turns (no moves), issued AFTER the step's weight loads, so
//     a wait for weights never covers a younger activation load;
//   * the residual lines of the whole wave tile are requested at the top of the tile's last K-step;
//   * ragged M: loads clamp the row to M - 1 (always a valid address), stores are masked.
// Same operand formats as conv_igemm_rb.hip: fp16 NHWC, weights in MFMA fragment order (ConvLaunch::w_frag), swapped operands
// (D = W * A^T), wave-private epilogue patches.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gdt_common.h"

#ifndef GDT_1X1_NT_RES
#define GDT_1X1_NT_RES 0       // 1: the residual (its last use in a ResNet block) is fetched non-temporal
#endif

namespace {

constexpr int ROWB = 128;                       // bytes per LDS row (64 halves of K)
constexpr int BM = 128, NT = 256;
constexpr int A_BYTES = BM * ROWB;              // 16 KB per stage
constexpr int PCP = 64 + 8;                     // patch row pitch in halves
constexpr int C_OFF = 2 * A_BYTES;
constexpr int BIAS_OFF = C_OFF + 4 * 32 * PCP * 2;
constexpr int MAX_COUT = 2048;                  // bias vector kept in LDS
constexpr size_t LDS_BYTES = (size_t)BIAS_OFF + MAX_COUT * 4;       // 59392: two workgroups per CU

struct TileAt { int tile_m, tile_n; bool valid; };
struct Pend { f16x8 v[4]; };

// TM = 2: waves 2 x 2, 64 x 64 each (tile 128 x 128); TM = 4: waves 1 x 4, 128 x 64 each (tile 128 x 256) -- a weight fragment
// then feeds four MFMAs: with two, the weight stream alone asks the full 64 B/clk of the CU's L1 at MFMA rate
// CAT: K-concatenated second operand (ConvLaunch::in2): K-steps [0, Cin / 64) read `in`, the rest read `in2` at the stride-in2_stride pixel of
// the output pixel (the row -> (n, oy, ox) split is done once per tile of the staging cursor)
// (the body takes its workgroup index and grid size as arguments: the multi-geometry entry below runs it per level, gdt_common.h MultiConv)
template <bool RES, int DEPTH, int TM, bool CAT = false>
__device__ __forceinline__ void conv1x1_rb_body(const ConvLaunch& d, const int vblocks, const int bid, const int gdim) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int WGN = TM == 4 ? 4 : 2, BN = WGN * 64, WTM = TM * 32;
    static_assert(TM == 2 || (TM == 4 && !RES), "wave tile");
    const int wm = wave / WGN, wn = wave % WGN;
    const int fr = lane & 31, fh = lane >> 5;

    const int ntm = (d.M + BM - 1) / BM, ntn = d.CoutPad / BN;
    auto tile_at = [&](int vb) -> TileAt {
        TileAt t;
        t.valid = vb < vblocks && gdt_tile_of_block(vb, ntm, ntn, t.tile_m, t.tile_n);
        if (!t.valid) { t.tile_m = 0; t.tile_n = 0; }
        else if (d.dbg & 2) t.tile_m = ntm - 1 - t.tile_m;        // rows from the end (see gdt_launch_conv_1x1_rb)
        return t;
    };
    int vb = bid;
    TileAt cur = tile_at(vb);
    if (!cur.valid) return;                   // (validity is monotone in vb)
    const int nk = d.Kpad >> 6, nks = d.Kpad >> 4;

    // ---- activation staging cursor: (tile, K-step) of the next piece set to load
    const int lrow = tid >> 3;                                 // 0..31; this thread's rows are lrow + 32 r
    const int q = (lane & 7) ^ ((lrow >> 1) & 7);              // source chunk of its 16-byte piece (XOR swizzle; 32 r keeps it)
    int s_vb = vb, s_tile_m = cur.tile_m, s_step = 0;
    const int nk1 = CAT ? d.Cin >> 6 : nk;                     // K-steps of the first operand
    unsigned row2[CAT ? 4 : 1];                                // element offsets of this thread's four rows in the second operand
    auto set_rows2 = [&]() {
        if (!CAT) return;
        const int hw = d.OH * d.OW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = min(s_tile_m * BM + r * 32 + lrow, d.M - 1);
            const int n_ = m / hw, rem = m - n_ * hw, oy = rem / d.OW, ox = rem - oy * d.OW;
            row2[CAT ? r : 0] = (unsigned)((n_ * d.in2_h + oy * d.in2_stride) * d.in2_w + ox * d.in2_stride) * (unsigned)d.in2_cin;
        }
    };
    set_rows2();
    auto load_pend = [&]() -> Pend {
        Pend p;
        const bool second = CAT && s_step >= nk1;              // (uniform)
        const f16* base = second ? d.in2 : d.in;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = min(s_tile_m * BM + r * 32 + lrow, d.M - 1);
            const unsigned off = second ? row2[CAT ? r : 0] + (unsigned)((s_step - nk1) * 64 + q * 8)
                                        : (unsigned)m * (unsigned)d.Cin + (unsigned)(s_step * 64 + q * 8);
            p.v[r] = *(const f16x8*)(base + off);
        }
        return p;
    };
    auto store_pend = [&](const Pend& p, int stage_off) {
#pragma unroll
        for (int r = 0; r < 4; ++r) *(f16x8*)(smem + stage_off + (r * 32 + lrow) * ROWB + ((lane & 7) << 4)) = p.v[r];
    };
    // past the end of a tile the cursor moves to the next tile of this workgroup (or parks on the current one: harmless re-reads)
    auto advance = [&]() {
        if (++s_step == nk) {
            s_step = 0;
            const TileAt nx = tile_at(s_vb + gdim);
            if (nx.valid) { s_tile_m = nx.tile_m; s_vb += gdim; set_rows2(); }
        }
    };

    // ---- weights: fragments straight from the fragment-ordered copy (uniform base + lane * 16 bytes)
    const unsigned lane_off = lane * 8;
    f16x8 b[4][2];
    auto load_b = [&](int kk, int tile_n, int step) {
        const f16* wb = d.w_frag + ((long)((tile_n * BN + wn * 64) / 32) * nks + step * 4) * 512;      // uniform
#pragma unroll
        for (int j = 0; j < 2; ++j) b[kk][j] = *(const f16x8*)(wb + ((long)j * nks * 512 + kk * 512) + lane_off);
    };
    const int a_lane = (wm * WTM + fr) * ROWB + ((fh ^ (((wm * WTM + fr) >> 1) & 7)) << 4);
    auto a_frag = [&](int stage_off, int i, int kk) -> f16x8 {
        return *(const f16x8*)(smem + ((a_lane + stage_off) ^ (kk << 5)) + i * 32 * ROWB);
    };

    // ---- prologue
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) load_b(kk, cur.tile_n, 0);
    for (int i = tid; i < d.CoutPad / 4; i += NT)
        *(float4*)(smem + BIAS_OFF + i * 16) = d.bias ? *(const float4*)(d.bias + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    { const Pend p0 = load_pend(); store_pend(p0, 0); advance(); }
    Pend P[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) { P[k] = load_pend(); advance(); }
    __syncthreads();
    f16x8 afr[2][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) afr[0][i] = a_frag(0, i, 0);

    f16* patch = (f16*)(smem + C_OFF) + wave * (32 * PCP);
    int so = 0;
    for (;;) {
        const TileAt nxt = tile_at(vb + gdim);
        f32x16 acc[TM][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        // residual lines / output offsets of this wave's 64 x 64 slice: row block i, line q -> pixel, 8 channels of lane & 7
        const int ch = lane & 7;
        const int col = cur.tile_n * BN + wn * 64 + ch * 8;
        f16x8 rv[RES ? TM : 1][4];
        auto out_off = [&](int i, int qq, bool& ok) -> unsigned {
            const int m = cur.tile_m * BM + wm * WTM + i * 32 + (lane >> 3) + 8 * qq;
            ok = (m < d.M) & (col < d.Cout);
            return ok ? (unsigned)m * (unsigned)d.Cout + (unsigned)col : 0u;        // offset 0 is a valid address for masked pieces
        };

        if (RES) {                                    // requested a whole K-loop ahead of their use
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    bool ok; const unsigned o = out_off(i, qq, ok);
#if GDT_1X1_NT_RES
                    typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
                    rv[RES ? i : 0][qq] = __builtin_bit_cast(f16x8, __builtin_nontemporal_load((const nt_u32x4*)(d.res + o)));
#else
                    rv[RES ? i : 0][qq] = *(const f16x8*)(d.res + o);
#endif
                }
        }
        auto k_step = [&](Pend& P, const int s, auto last_tag) {
            constexpr bool last = decltype(last_tag)::value;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int cu = kk & 1, nx = cu ^ 1;
                if (kk < 3) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) afr[nx][i] = a_frag(so, i, kk + 1);
                }
                if (kk == 1) store_pend(P, A_BYTES - so);                 // activations of the next step
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[kk][j], afr[cu][i], acc[i][j], 0, 0, 0);   // D[cout][pixel]
                if (!last) load_b(kk, cur.tile_n, s + 1);
                else load_b(kk, nxt.tile_n, 0);                           // the next tile's first slice: ahead of the epilogue's stores
                if (kk == 3) { P = load_pend(); advance(); }              // ... of the step after it (DEEP: two after), behind the weights
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            so = A_BYTES - so;
            if (!last) {
#pragma unroll
                for (int i = 0; i < TM; ++i) afr[0][i] = a_frag(so, i, 0);
            }
        };
        // nk % DEPTH == 0: the register sets take turns (static indices, no moves)
        for (int s = 0; s + DEPTH < nk; s += DEPTH) {
#pragma unroll
            for (int k = 0; k < DEPTH; ++k) k_step(P[k], s + k, std::false_type());
        }
#pragma unroll
        for (int k = 0; k + 1 < DEPTH; ++k) k_step(P[k], nk - DEPTH + k, std::false_type());
        k_step(P[DEPTH - 1], nk - 1, std::true_type());

        // ------------------------------------------------------------ wave-private epilogue (as conv_igemm_rb.hip): lane (fr, fh) holds
        // pixel fr of row block i and, in registers 4g .. 4g+3, the output channels 8g + 4fh .. +3 of column block j.  Per row block
        // the wave transposes its 32 x 64 slice through its own LDS patch (8-byte writes, 16-byte reads) and stores one 128-byte
        // line per pixel; bias, residual, ReLU on the way.  No workgroup barrier.
        {
            const bool relu_now = d.relu && !RES;
            float4 bvs[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    bvs[j][g] = *(const float4*)(smem + BIAS_OFF + (cur.tile_n * BN + wn * 64 + j * 32 + 8 * g + 4 * fh) * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bv = bvs[j][g];
                        const f32x16& a = acc[i][j];
                        float v0 = a[4 * g] + bv.x, v1 = a[4 * g + 1] + bv.y, v2 = a[4 * g + 2] + bv.z, v3 = a[4 * g + 3] + bv.w;
                        if (relu_now) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        f16x4 h; h[0] = (f16)v0; h[1] = (f16)v1; h[2] = (f16)v2; h[3] = (f16)v3;
                        *(f16x4*)(patch + fr * PCP + j * 32 + 8 * g + 4 * fh) = h;
                    }
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    f16x8 v = *(const f16x8*)(patch + ((lane >> 3) + 8 * qq) * PCP + ch * 8);
                    if (RES) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float t = (float)v[e] + (float)rv[RES ? i : 0][qq][e];
                            if (d.relu) t = fmaxf(t, 0.f);
                            v[e] = (f16)t;
                        }
                    }
                    bool ok; const unsigned o = out_off(i, qq, ok);
                    if (ok) *(f16x8*)(d.out + o) = v;
                }
            }
        }
        if (!nxt.valid) break;
        cur = nxt; vb += gdim;
#pragma unroll
        for (int i = 0; i < TM; ++i) afr[0][i] = a_frag(so, i, 0);
    }
}

template <bool RES, int DEPTH, int TM, bool CAT = false>
__global__ __launch_bounds__(NT, 2) void conv1x1_rb_kernel(const ConvLaunch d, const int vblocks) {
    conv1x1_rb_body<RES, DEPTH, TM, CAT>(d, vblocks, blockIdx.x, gridDim.x);
}
template <bool RES, int DEPTH, int TM, bool CAT = false>
__global__ __launch_bounds__(NT, 2) void conv1x1_rb_multi_kernel(const MultiConv m) {
    const int l = gdt_multi_level(m.nlev, m.prefix, blockIdx.x);
    conv1x1_rb_body<RES, DEPTH, TM, CAT>(m.lev[l], m.vblocks[l], blockIdx.x - m.prefix[l], m.prefix[l + 1] - m.prefix[l]);
}

template <bool RES, int DEPTH, int TM, bool CAT = false>
int launch_1x1(const ConvLaunch* dl, int L, hipStream_t stream) {
    const ConvLaunch& d = dl[0];
    static GdtPerDevice per_dev;          // (hipFuncSetAttribute is per device: gdt_common.h)
    int slots = 0;
    {
        const int rc = gdt_per_device(per_dev, slots, [](int, int cus, int& v) {
            int per_cu = 0;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv1x1_rb_kernel<RES, DEPTH, TM, CAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv1x1_rb_multi_kernel<RES, DEPTH, TM, CAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
            GDT_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv1x1_rb_kernel<RES, DEPTH, TM, CAT>, NT, LDS_BYTES));
            static const int cap = [] { const char* e = getenv("GDT_CONV_1X1_WPC"); return e ? atoi(e) : 3; }();
            if (per_cu < 1) per_cu = 1;
            if (per_cu > cap) per_cu = cap;
            v = cus / 8 * 8 * per_cu;                 // a multiple of 8: a workgroup's tiles stay on its XCD
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    if (L > 1) {
        MultiConv m;
        m.nlev = L;
        for (int l = 0; l < L; ++l) { m.lev[l] = dl[l]; m.vblocks[l] = gdt_grid_for_tiles((dl[l].M + BM - 1) / BM, dl[l].CoutPad / (TM == 4 ? 256 : 128)); }
        const int grid = gdt_multi_partition(m.prefix, m.vblocks, L, slots);
        hipLaunchKernelGGL((conv1x1_rb_multi_kernel<RES, DEPTH, TM, CAT>), dim3(grid), dim3(NT), LDS_BYTES, stream, m);
        GDT_CHECK_HIP(hipGetLastError());
        return GDT_OK;
    }
    const int vblocks = gdt_grid_for_tiles((d.M + BM - 1) / BM, d.CoutPad / (TM == 4 ? 256 : 128));
    const int grid = vblocks < slots ? vblocks : slots;
    hipLaunchKernelGGL((conv1x1_rb_kernel<RES, DEPTH, TM, CAT>), dim3(grid), dim3(NT), LDS_BYTES, stream, d, vblocks);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// Eligible: a dense 1x1 stride-1 conv in fp16 NHWC with fragment-ordered weights, Cin % 64 == 0, Cout in whole 128-column tiles,
// an even number of K-steps (or a single one), no fused statistics / input transform, 32-bit element offsets, enough tiles.
bool gdt_conv_1x1_rb_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_1X1"); return e ? atoi(e) : 1; }();   // 0 off, 2 force
    if (mode == 0 || !d.w_frag || d.out_f32 || !d.out) return false;
    if (d.ntaps != 1 || d.sy != 1 || d.sx != 1 || d.dy0 != 0 || d.dx0 != 0 || d.osy != 1 || d.osx != 1 || d.ooy != 0 || d.oox != 0) return false;
    if (d.OHg != d.OH || d.OWg != d.OW || d.OH != d.H || d.OW != d.W) return false;
    if (d.Cin % 64 != 0 || d.Kpad != d.Cin || d.CoutPad % 128 != 0 || d.CoutPad > MAX_COUT || d.Cout % 8 != 0) return false;
    const int nk = d.Kpad / 64;
    if (nk != 1 && nk % 2 != 0) return false;
    if (d.stats || d.in_norm || d.in_res || d.in_out || d.phase_cout || d.pool2) return false;
    if ((long)d.M * d.Cin >= (1L << 32) || (long)d.M * d.Cout >= (1L << 32)) return false;
    if (mode == 2) return true;
    static const int min_tiles = [] { const char* e = getenv("GDT_CONV_1X1_MIN_TILES"); return e ? atoi(e) : 64; }();     // (batch 2-4 @1024^2: +7 % over 512; multi-scale config +3 %)
    return (long)((d.M + BM - 1) / BM) * (d.CoutPad / 128) >= min_tiles;
}

// K-concatenated form: both operands in whole 64-channel K-steps, an even number of them in total, 256-wide tiles, no residual read (the shortcut
// IS the second operand), 32-bit element offsets into both inputs
bool gdt_conv_1x1_cat_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_1X1_CAT"); return e ? atoi(e) : 1; }();   // 0 off
    if (mode == 0 || !d.in2 || !d.w_frag || d.out_f32 || !d.out || d.res) return false;
    if (d.ntaps != 1 || d.sy != 1 || d.sx != 1 || d.dy0 != 0 || d.dx0 != 0 || d.osy != 1 || d.osx != 1 || d.ooy != 0 || d.oox != 0) return false;
    if (d.OHg != d.OH || d.OWg != d.OW || d.OH != d.H || d.OW != d.W) return false;
    if (d.Cin % 64 != 0 || d.in2_cin % 64 != 0 || d.Kpad != d.Cin + d.in2_cin || (d.Kpad / 64) % 2 != 0 || d.CoutPad % 256 != 0 || d.CoutPad > MAX_COUT || d.Cout % 8 != 0) return false;
    if (d.in2_stride < 1 || d.in2_stride > 2 || (d.OH - 1) * d.in2_stride >= d.in2_h || (d.OW - 1) * d.in2_stride >= d.in2_w) return false;
    if (d.stats || d.in_norm || d.in_res || d.in_out || d.phase_cout || d.pool2) return false;
    if ((long)d.M * d.Cin >= (1L << 32) || (long)d.M * d.Cout >= (1L << 32) || (long)d.N * d.in2_h * d.in2_w * d.in2_cin >= (1L << 32)) return false;
    return (long)((d.M + BM - 1) / BM) * (d.CoutPad / 128) >= 64;
}

// `dl[0 .. L)`: the same conv on L independent geometries (the levels of a pyramid) as ONE launch; the instantiation is chosen for the levels together
int gdt_launch_conv_1x1_rb_levels(const ConvLaunch* dl_in, int L, hipStream_t stream) {
    // Row order: the reduce convs walk their rows from the END.  Their input is what the expand conv of the previous block has just
    // written front to back, so the rows written last -- the ones still in the 256 MB Infinity Cache -- are read first (and the 3x3
    // conv that follows, front to back, starts on the rows THIS launch wrote last).  ResNet-101 batch 32: 1860 -> 1882 descriptors/s;
    // reversing the expand convs instead gives the same, reversing both nothing (GDT_CONV_1X1_REV: 1 reduce, 2 expand, 3 both, 0 none).
    static const int rev = [] { const char* e = getenv("GDT_CONV_1X1_REV"); return e ? atoi(e) : 1; }();
    GDT_REQUIRE(L >= 1 && L <= GDT_MAX_LEVELS, "1..4 geometries per launch");
    ConvLaunch dl[GDT_MAX_LEVELS];
    long wide_tiles = 0;
    for (int l = 0; l < L; ++l) {
        dl[l] = dl_in[l];
        dl[l].dbg = ((rev & 1) && !dl[l].res) || ((rev & 2) && dl[l].res) ? 2 : 0;
        GDT_REQUIRE(dl[l].Kpad == dl[0].Kpad && dl[l].CoutPad == dl[0].CoutPad && (dl[l].res != nullptr) == (dl[0].res != nullptr) && (dl[l].in2 != nullptr) == (dl[0].in2 != nullptr),
                    "the geometries of one launch run the same conv");
        wide_tiles += (long)((dl[l].M + BM - 1) / BM) * (dl[l].CoutPad / 256);
    }
    const ConvLaunch& d = dl[0];
    if (d.in2) return launch_1x1<false, 2, 4, true>(dl, L, stream);
    const int nk = d.Kpad / 64;
    static const int max_depth = [] { const char* e = getenv("GDT_CONV_1X1_DEPTH"); return e ? atoi(e) : 4; }();
    static const int wide = [] { const char* e = getenv("GDT_CONV_1X1_WIDE"); return e ? atoi(e) : 1; }();
    if (d.res) return nk > 1 ? launch_1x1<true, 2, 2>(dl, L, stream) : launch_1x1<true, 1, 2>(dl, L, stream);
    // (the 128 x 256 tile halves the weight stream per MFMA but also the number of workgroups: below ~one per CU the 128 x 128 tile fills the chip better)
    static const int wide_min = [] { const char* e = getenv("GDT_CONV_1X1_WIDE_MIN"); return e ? atoi(e) : 256; }();
    if (wide && nk > 1 && d.CoutPad % 256 == 0 && wide_tiles >= wide_min) return launch_1x1<false, 2, 4>(dl, L, stream);
    if (nk % 4 == 0 && max_depth >= 4) return launch_1x1<false, 4, 2>(dl, L, stream);
    return nk > 1 ? launch_1x1<false, 2, 2>(dl, L, stream) : launch_1x1<false, 1, 2>(dl, L, stream);
}

int gdt_launch_conv_1x1_rb(const ConvLaunch& d, hipStream_t stream) { return gdt_launch_conv_1x1_rb_levels(&d, 1, stream); }
