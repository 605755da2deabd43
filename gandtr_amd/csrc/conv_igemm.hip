// Implicit-GEMM convolution for gfx950 (MI355X): fp16 NHWC activations x fp16 packed weights -> fp32 MFMA accumulate.
//
// Replaces every nn.Conv2d / nn.ConvTranspose2d on the gandtr hot path:
//   generator  mdir/components/model/network/p2p_networks.py:269-311, ResnetBlock :480-494
//   embedders  torchvision VGG16 / ResNet-101 trunks sliced at external/cirtorch/networks/imageretrievalnet.py:185-190
//   HED        mdir/components/model/network/hed.py:50-58
//
// GEMM view: M = N*OHg*OWg output positions, N = Cout, K = taps*Cin (k = tap*Cin + c).
//   A[m][k] = in[n][oy*sy+dy(tap)][ox*sx+dx(tap)][c]  (zero or reflect padding resolved per 16-byte chunk)
//   B[k][n] = w[n][k]
// Tile BM x BN x 64 per workgroup, 4 wavefronts (64 lanes each), v_mfma_f32_32x32x16_f16.
// Both operands are staged global -> LDS with global_load_lds_dwordx4 (no VGPR round trip); the LDS image is
// linear per wave-instruction, so the bank-conflict XOR swizzle is applied on the SOURCE chunk index and again on
// the ds_read_b128 address (chunk' = chunk ^ ((row >> 1) & 7)), which is conflict-free for the 32x32x16 fragment
// reads.  Two LDS stages, one barrier per K-step: the loads of step k+1 are in flight while step k computes.
#include <cstdlib>

#include "conv_epilogue.h"
#include "gdt_common.h"

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int BK = 64;           // K-step (halves) = 128 B per tile row
constexpr int ROWB = BK * 2;     // bytes per LDS tile row

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_dst, 16, 0, 0);
}

template <int BM, int BN, int WGM, int WGN, bool NORM = false>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_igemm_kernel(const ConvLaunch d) {
    constexpr int NT = WGM * WGN * 64;                // threads per workgroup (4 or 8 wavefronts)
    constexpr int RPR = NT / 8;                       // tile rows staged per loader round (8 lanes x 16 B per row)
    constexpr int WTM = BM / WGM, WTN = BN / WGN;     // per-wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;       // 32x32 MFMA tiles per wave
    constexpr int AR = BM / RPR, BR = BN / RPR;       // loader rounds
    static_assert(BM % RPR == 0 && BN % RPR == 0 && RPR % 16 == 0, "tile / workgroup mismatch");
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    const int ntn = d.CoutPad / BN;
    int tile_m, tile_n;
    if (!gdt_tile_of_block(blockIdx.x, (d.M + BM - 1) / BM, ntn, tile_m, tile_n)) return;      // XCD-chunked, see gdt_common.h

    // ---- per-thread loader state: this thread stages 16-byte chunk (lane & 7) of rows r = round*RPR + wave*8 + lane/8
    const int lrow = wave * 8 + (lane >> 3);
    const int swz = (lrow >> 1) & 7;                  // same for every round (round*RPR >> 1 is a multiple of 8)
    const int q = (lane & 7) ^ swz;                   // source chunk (8 halves of K) this lane fetches
    const int hw_g = d.OHg * d.OWg;
    int a_base[AR];                                   // pixel index of image n's first pixel (N*H*W < 2^31)
    int a_iy0[AR], a_ix0[AR];
    unsigned a_valid = 0;                             // bit r: row of round r is inside M
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const int m = tile_m * BM + r * RPR + lrow;
        const int mm = m < d.M ? m : 0;
        const int n = mm / hw_g, rem = mm - n * hw_g;
        const int oy = rem / d.OWg, ox = rem - oy * d.OWg;
        a_base[r] = n * d.H * d.W;
        a_iy0[r] = oy * d.sy;
        a_ix0[r] = ox * d.sx;
        a_valid |= (m < d.M ? 1u : 0u) << r;
    }
    const f16* b_src = d.w + ((long)(tile_n * BN + lrow) * d.Kpad + q * 8);
    const int cmask = (1 << d.lc8) - 1;

    // Staging of one K-step is cut into per-round pieces so that the main loop can spread the global_load_lds
    // instructions (and their address arithmetic) between the MFMAs of the current step instead of issuing them in a burst.
    int n_dy = 0, n_dx = 0, n_c8 = 0; bool n_tap_ok = false;      // decoded (tap, channel chunk) of the NEXT K-step
    const bool refl = d.pad_reflect != 0;
    auto decode = [&](int ks) {
        const int k8 = ks * 8 + q;
        const int tap = k8 >> d.lc8;
        n_c8 = k8 & cmask;
        const int ty = (tap * d.invTW) >> 16, tx = tap - ty * d.TW;
        n_dy = d.dy0 + ty * d.dys; n_dx = d.dx0 + tx * d.dxs;
        n_tap_ok = tap < d.ntaps;
    };
    auto issue_a = [&](int stage, int r) {
        char* As = smem + stage * A_BYTES;
        // branch-free padding: compute the reflected index and the in-bounds predicate, select afterwards
        const int iy = a_iy0[r] + n_dy, ix = a_ix0[r] + n_dx;
        const int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
        const int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
        const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
        const bool ok = n_tap_ok & (((a_valid >> r) & 1u) != 0) & (inb | refl);
        const int pix = a_base[r] + ry * d.W + rx;           // reflected == identity when in bounds
        const f16* src = d.in + (((long)pix << (d.lc8 + 3)) + n_c8 * 8);
        glds16(ok ? src : d.zeros, As + (r * RPR + wave * 8) * ROWB);
    };
    // Fused InstanceNorm(+ReLU) of the producer for 64-channel inputs (generator: stem -> first down conv, last up conv -> head;
    // p2p_networks.py:271-272, :299-300): the A operand goes through registers -- raw fp16 chunk, x -> max((x-mean)*rstd, 0)
    // in fp32, ds_write_b128 into the slot the DMA path would have filled.  With Cin == 64 a lane's channel group (q) is
    // the same in every K-step, and the tile lies inside one image, so its 8 (mean, rstd) pairs are loaded once.
    constexpr bool norm_a = NORM;                       // separate instantiation: keeps the DMA-only form lean
    float nmr[16];
    if constexpr (norm_a) {
        const int n_img = (tile_m * BM) / hw_g;
        const float4* p4 = (const float4*)(d.in_norm + ((long)n_img * d.Cin + q * 8) * 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float4 v = p4[k]; nmr[4 * k] = v.x; nmr[4 * k + 1] = v.y; nmr[4 * k + 2] = v.z; nmr[4 * k + 3] = v.w; }
    }
    f16x8 areg[AR]; unsigned areg_ok = 0;
    auto load_a_regs = [&]() {                          // uses the decoded (tap, chunk) of the NEXT K-step
        areg_ok = 0;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int iy = a_iy0[r] + n_dy, ix = a_ix0[r] + n_dx;
            const int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
            const int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
            const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
            const bool ok = n_tap_ok & (((a_valid >> r) & 1u) != 0) & (inb | refl);
            const int pix = a_base[r] + ry * d.W + rx;
            const f16* src = ok ? d.in + (((long)pix << (d.lc8 + 3)) + n_c8 * 8) : d.zeros;
            areg[r] = *(const f16x8*)src;
            areg_ok |= (ok ? 1u : 0u) << r;
        }
    };
    auto store_a_regs = [&](int stage) {
        char* As = smem + stage * A_BYTES;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            f16x8 o;
            const bool ok = (areg_ok >> r) & 1u;        // padded / out-of-range positions stay exactly zero
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = ((float)areg[r][e] - nmr[2 * e]) * nmr[2 * e + 1];
                if (d.in_relu) f = fmaxf(f, 0.f);
                o[e] = ok ? (f16)f : (f16)0.f;
            }
            *(f16x8*)(As + (r * RPR + lrow) * ROWB + ((lane & 7) << 4)) = o;
        }
    };
    auto issue_b = [&](int ks, int stage, int r) {
        char* Bs = smem + 2 * A_BYTES + stage * B_BYTES;
        glds16(b_src + ((long)r * RPR * d.Kpad + ks * BK), Bs + (r * RPR + wave * 8) * ROWB);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment read addresses (row part); chunk part is XORed per k16 step
    const int fr = lane & 31, fh = lane >> 5;
    int a_off[TM], a_sw[TM], b_off[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + fr;
        a_off[i] = row * ROWB; a_sw[i] = (row >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + fr;
        b_off[j] = row * ROWB; b_sw[j] = (row >> 1) & 7;
    }

    decode(0);
    if constexpr (norm_a) { load_a_regs(); store_a_regs(0); }
    else {
#pragma unroll
        for (int r = 0; r < AR; ++r) issue_a(0, r);
    }
#pragma unroll
    for (int r = 0; r < BR; ++r) issue_b(0, 0, r);

    constexpr int KK = BK / 16;                       // MFMA k-substeps per K-step
    constexpr int APK = (AR + KK - 1) / KK, BPK = (BR + KK - 1) / KK;   // staging rounds issued per k-substep
    f16x8 afr[2][TM], bfr[2][TN];                     // fragment double buffer (statically indexed after unrolling)
    for (int ks = 0; ks < d.nk; ++ks) {
        __syncthreads();   // stage ks landed (vmcnt(0) precedes the barrier); everyone is done reading the other stage
        const bool more = (ks + 1 < d.nk) && !(d.dbg & 1);
        const int nst = (ks + 1) & 1;
        if (more) decode(ks + 1);
        if constexpr (norm_a) { if (more) load_a_regs(); }   // issue-early; normalised and written after this step's MFMAs
        const char* As = smem + (ks & 1) * A_BYTES;
        const char* Bs = smem + 2 * A_BYTES + (ks & 1) * B_BYTES;
        if (d.dbg & 2) {
            if (more) {
#pragma unroll
                for (int r = 0; r < AR; ++r) issue_a(nst, r);
#pragma unroll
                for (int r = 0; r < BR; ++r) issue_b(ks + 1, nst, r);
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) afr[0][i] = *(const f16x8*)(As + a_off[i] + ((fh ^ a_sw[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[0][j] = *(const f16x8*)(Bs + b_off[j] + ((fh ^ b_sw[j]) << 4));
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KK) {            // fragments of the next k-substep: in flight while this one's MFMAs run
                const int ch = 2 * (kk + 1) + fh;
#pragma unroll
                for (int i = 0; i < TM; ++i) afr[nxt][i] = *(const f16x8*)(As + a_off[i] + ((ch ^ a_sw[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[nxt][j] = *(const f16x8*)(Bs + b_off[j] + ((ch ^ b_sw[j]) << 4));
            }
            if (!norm_a && more) {
#pragma unroll
                for (int r = kk * APK; r < (kk + 1) * APK && r < AR; ++r) issue_a(nst, r);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
                    if (more && i * TN + j == ((TM * TN) / 2 > 0 ? (TM * TN) / 2 - 1 : 0)) {
#pragma unroll
                        for (int r = kk * BPK; r < (kk + 1) * BPK && r < BR; ++r) issue_b(ks + 1, nst, r);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (norm_a) { if (more) store_a_regs(nst); }
    }

    // ---------------------------------------------------------------- epilogue
    const int ohw = d.OH * d.OW;
    if (d.out_f32) {
        // fp32 NCHW output straight from the accumulators (generator head: Cout = 3, + bias, tanh)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = tile_n * BN + wn * WTN + j * 32 + fr;
            if (col >= d.Cout) continue;
            const float bv = d.bias ? d.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    const int m = tile_m * BM + row;
                    if (m >= d.M) continue;
                    const int n = m / hw_g, rem = m - n * hw_g;
                    const int oy = rem / d.OWg, ox = rem - oy * d.OWg;
                    float v = acc[i][j][e] + bv;
                    if (d.relu) v = fmaxf(v, 0.f);
                    if (d.act == 1) v = tanhf(v);
                    else if (d.act == 2) v = 1.f / (1.f + __expf(-v));
                    d.out_f32[((long)n * d.Cout + col) * ohw + (long)(oy * d.osy + d.ooy) * d.OW + ox * d.osx + d.oox] = v;
                }
        }
        return;
    }

    // fp16 NHWC output (conv_epilogue.h): LDS transpose, fused InstanceNorm statistics, residual add
    conv_epilogue_f16<BM, BN, WGM, WGN, NT, TM, TN>(d, acc, smem, tile_m, tile_n, [&](int row, bool& ok) -> long {
        const int m = tile_m * BM + row;
        ok = m < d.M;
        const int mm = ok ? m : 0;
        const int n = mm / hw_g, rem = mm - n * hw_g;
        const int oy = rem / d.OWg, ox = rem - oy * d.OWg;
        return (long)n * ohw + (long)(oy * d.osy + d.ooy) * d.OW + ox * d.osx + d.oox;
    });
}

template <int BM, int BN, int WGM, int WGN>
constexpr size_t lds_bytes() {
    constexpr size_t staging = 2 * (size_t)(BM + BN) * ROWB;
    constexpr size_t epilogue = conv_epilogue_lds_bytes<BM, BN, WGM, WGN, WGM * WGN * 64>();
    return staging > epilogue ? staging : epilogue;
}

template <int BM, int BN, int WGM, int WGN, bool NORM = false>
int launch_cfg(const ConvLaunch& d, hipStream_t stream) {
    const int ntm = (d.M + BM - 1) / BM, ntn = d.CoutPad / BN;
    constexpr size_t lds = lds_bytes<BM, BN, WGM, WGN>();
    static_assert(lds <= 160 * 1024, "LDS budget (160 KiB per CU on gfx950)");
    if (lds > 64 * 1024) {
        static GdtPerDevice per_dev;          // one attribute call per template instantiation AND device (gdt_common.h)
        int attr_set = 0;
        {
            const int rc = gdt_per_device(per_dev, attr_set, [](int, int, int& v) {
                v = 1;
                GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WGM, WGN, NORM>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                return GDT_OK;
            });
            if (rc != GDT_OK) return rc;
        }
    }
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, NORM>), dim3(gdt_grid_for_tiles(ntm, ntn)), dim3(WGM * WGN * 64), lds, stream, d);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// the generic kernel can apply the producer's InstanceNorm while staging when a lane's channel group is step-invariant
// (Cin == 64) and a tile never straddles two images
bool gdt_conv_igemm_norm_eligible(const ConvLaunch& d) {
    return d.Cin == 64 && (d.OHg * d.OWg) % 256 == 0 && d.M % 256 == 0;
}

// A fused 2x2 max pool needs a patch kernel (16x16 patches: conv3x3_halo_rb.hip / conv3x3_halo.hip) and a plain
// epilogue (no statistics, residual or write-back).
bool gdt_conv_pool2_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_POOL"); return e ? atoi(e) : 1; }();
    if (mode == 0 || d.H < 2 || d.W < 2 || d.stats || d.res || d.in_out || d.out_f32 || d.phase_cout) return false;
    return gdt_conv_halo_rb_eligible(d) || gdt_conv_halo_eligible(d);
}

int gdt_conv_bn(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }

// Which kernel family gdt_launch_conv picks for `d` (same order of choice), for the families that can run several geometries in one launch:
// 1 = conv1x1_rb.hip, 2 = conv3x3_halo_rb.hip, 0 = any other
int gdt_conv_family(const ConvLaunch& d) {
    if (d.pool2 || d.stats || d.in_norm) return 0;
    if (gdt_conv_stem_eligible(d)) return 0;
    if (gdt_conv_1x1_rb_eligible(d)) return 1;
    if (!gdt_conv_halo_rb_eligible(d) && !gdt_conv_halo_eligible(d) && gdt_conv_igemm_rb_eligible(d)) return 0;
    if (gdt_conv_halo_rb_eligible(d)) return 2;
    return 0;
}

int gdt_launch_conv(const ConvLaunch& d_in, hipStream_t stream, int* variant) {
    int vdummy; if (!variant) variant = &vdummy;
    static const int dbg = [] { const char* e = getenv("GDT_CONV_DBG"); return e ? atoi(e) : 0; }();
    ConvLaunch d = d_in;
    d.dbg = dbg;
    GDT_REQUIRE(d.Cin >= 8 && (d.Cin & (d.Cin - 1)) == 0, "Cin must be a power of two >= 8");
    GDT_REQUIRE((1 << d.lc8) * 8 == d.Cin, "lc8 mismatch");
    GDT_REQUIRE(d.Kpad % BK == 0 && d.nk == d.Kpad / BK && d.nk >= 1, "Kpad must be a multiple of 64");
    GDT_REQUIRE(d.ntaps * d.Cin <= d.Kpad, "Kpad too small");
    GDT_REQUIRE(d.out_f32 || (d.Cout % 8 == 0), "fp16 NHWC output needs Cout % 8 == 0");
    GDT_REQUIRE(d.M > 0 && d.zeros != nullptr, "empty launch");
    if (d.stats) GDT_REQUIRE(!d.out_f32 && !d.res && !d.relu && d.M % 128 == 0 && (d.OHg * d.OWg) % 128 == 0,
                             "fused InstanceNorm statistics need whole 128-row tiles per image and a plain conv epilogue");
    const int bn = gdt_conv_bn(d.Cout);
    GDT_REQUIRE(d.CoutPad % bn == 0 && d.CoutPad >= d.Cout, "CoutPad must be a multiple of the N tile");
    if (d.pool2) GDT_REQUIRE(gdt_conv_halo_rb_eligible(d) || gdt_conv_halo_eligible(d), "fused max pool needs a patch kernel");
    if (gdt_conv_stem_eligible(d)) { *variant = 950000 + d.ntaps; return gdt_launch_conv_stem(d, stream); }
    if (gdt_conv_1x1_rb_eligible(d)) { *variant = 945128; return gdt_launch_conv_1x1_rb(d, stream); }
    if (!gdt_conv_halo_rb_eligible(d) && !gdt_conv_halo_eligible(d) && gdt_conv_igemm_rb_eligible(d)) return gdt_launch_conv_igemm_rb(d, stream, variant);
    if (d.in_norm && !gdt_conv_halo_eligible(d))
        GDT_REQUIRE(gdt_conv_igemm_norm_eligible(d), "fused input normalisation needs Cin == 64 and whole 256-row tiles per image here");
    if (d.in_norm && !gdt_conv_halo_eligible(d)) {      // generic kernel with the producer's InstanceNorm folded into the A staging
        if (bn == 128) { *variant = 256128; return launch_cfg<256, 128, 4, 2, true>(d, stream); }
        if (bn == 64) { *variant = 128064; return launch_cfg<128, 64, 2, 2, true>(d, stream); }
        *variant = 128032; return launch_cfg<128, 32, 4, 1, true>(d, stream);
    }
    if (gdt_conv_halo_rb_eligible(d)) { *variant = 910000 + (d.CoutPad < 256 ? d.CoutPad : 256); return gdt_launch_conv_halo_rb(d, stream); }
    if (gdt_conv_halo_eligible(d)) { *variant = 900000 + (d.CoutPad % 256 == 0 ? 256 : (d.CoutPad % 128 == 0 ? 128 : 64)); return gdt_launch_conv_halo(d, stream); }
    if (d.pad_reflect) {
        const int pady = d.dy0 < 0 ? -d.dy0 : 0, padx = d.dx0 < 0 ? -d.dx0 : 0;
        GDT_REQUIRE(pady < d.H && padx < d.W, "reflect padding needs pad < input size");
    }
    // large problems: 256-row tiles with 8 wavefronts (half the L2->LDS bytes per FLOP of the 128x128 tile); the grid must
    // still cover the 256 CUs a few times over
    const long tiles256 = ((long)d.M + 255) / 256;
    static const int force_tile = [] { const char* e = getenv("GDT_CONV_TILE"); return e ? atoi(e) : 0; }();   // test knob
    const long min_blocks = force_tile == 256 ? 1 : 512;
    static const int min_nk = [] { const char* e = getenv("GDT_CONV_MINK"); return e ? atoi(e) : 0; }();
    if (force_tile != 128 && (d.nk >= min_nk || force_tile == 256)) {
        if (!d.in_norm && d.CoutPad % 256 == 0 && tiles256 * (d.CoutPad / 256) >= min_blocks) { *variant = 256256; return launch_cfg<256, 256, 2, 4>(d, stream); }
        if (bn == 128 && tiles256 * (d.CoutPad / 128) >= min_blocks) { *variant = 256128; return launch_cfg<256, 128, 4, 2>(d, stream); }
        if (bn == 64 && tiles256 * (d.CoutPad / 64) >= 2 * min_blocks) { *variant = 256064; return launch_cfg<256, 64, 4, 2>(d, stream); }
    }
    GDT_REQUIRE(d.in_norm == nullptr, "fused input normalisation is only implemented in the halo kernels");
    // under-filled launches (small batches): narrower N tiles put more workgroups on the chip; the K loop is what bounds them
    const long tiles128 = ((long)d.M + 127) / 128;
    if (bn == 128 && tiles128 * (d.CoutPad / 128) < 192 && d.CoutPad % 64 == 0 && force_tile == 0) {
        *variant = 128064;
        return launch_cfg<128, 64, 2, 2>(d, stream);
    }
    *variant = 128000 + bn;
    if (bn == 128) return launch_cfg<128, 128, 2, 2>(d, stream);
    if (bn == 64) return launch_cfg<128, 64, 2, 2>(d, stream);
    return launch_cfg<128, 32, 4, 1>(d, stream);
}
