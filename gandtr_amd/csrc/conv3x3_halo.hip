// 3x3 / stride-1 / pad-1 convolution with an LDS-resident input halo tile (gfx950, MI355X).
//
// Hot layers: the 18 ResnetBlock convs of the generator (p2p_networks.py:480-494; 88 % of its FLOPs), every VGG16 / HED
// conv but the first (torchvision cfg "D"; hed.py:50-58) and the stride-1 3x3 convs of the ResNet-101 Bottlenecks.
//
// The generic implicit GEMM (conv_igemm.hip) re-stages the A operand for each of the 9 taps, and measures out as bound by
// the per-CU L2->LDS rate (~50 GB/s), not by MFMA.  Here one workgroup owns a 16x16 output patch of one image (M = 256):
// per 64-channel chunk it stages the 18x18 input halo ONCE (41.5 KB, padding resolved in the per-lane source address) and
// runs the 9 taps against it by offsetting the fragment row (halo row = (py+ty)*18 + px+tx); only the weights (BN x 64 per
// tap) stream every step.  L2->LDS bytes per FLOP drop by ~43 %.
//   8 wavefronts, v_mfma_f32_32x32x16_f16, two LDS stages for the halo (per chunk) and for the weights (per tap),
//   one barrier per (chunk, tap) step, weight loads interleaved with the MFMAs, halo rounds spread over taps 0..5 of the
//   previous chunk.  Same XOR swizzle (chunk' = chunk ^ ((row >> 1) & 7)) and the same epilogue as conv_igemm.hip.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv_epilogue.h"
#include "gdt_common.h"

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int ROWB = 128;          // bytes per LDS row (64 halves of K)
constexpr int HALO_W = 18;
// patch height PH (16 or 8 output rows x 16 columns): halo (PH+2) x 18 rows of 128 B, padded to a multiple of 8 rows
constexpr int halo_rows(int PH) { return (PH + 2) * HALO_W; }
constexpr int halo_rows_pad(int PH) { return (halo_rows(PH) + 7) / 8 * 8; }
constexpr int a_bytes(int PH) { return halo_rows_pad(PH) * ROWB; }

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_dst, 16, 0, 0);
}

template <int PH, int BN, int WGM, int WGN>
constexpr size_t halo_lds_bytes() {
    constexpr size_t staging = 2 * (size_t)a_bytes(PH) + 2 * (size_t)BN * ROWB + 1024;   // + (mean, rstd) of two chunks
    constexpr size_t epilogue = conv_epilogue_lds_bytes<PH * 16, BN, WGM, WGN, WGM * WGN * 64>();
    return staging > epilogue ? staging : epilogue;
}

// PH = 16, 8 wavefronts: one 256 x BN tile per CU.  PH = 8, 4 wavefronts, BN = 128: 78 KB of LDS, so TWO workgroups share a
// CU and fill each other's barrier / first-fragment bubbles (the single-workgroup form spends ~30 % of its wave cycles there).
template <int PH, int BN, int WGM, int WGN>
__global__ __launch_bounds__(WGM * WGN * 64) void conv3x3_halo_kernel(const ConvLaunch d) {
    constexpr int NT = WGM * WGN * 64, RPR = NT / 8;   // threads, tile rows staged per loader round
    constexpr int BM = PH * 16;
    constexpr int HALO_ROWS = halo_rows(PH), HALO_ROWS_PAD = halo_rows_pad(PH), A_BYTES = a_bytes(PH);
    constexpr int NR = (HALO_ROWS_PAD + RPR - 1) / RPR;         // halo staging rounds per chunk
    static_assert(NR <= 6, "halo rounds are spread over taps 0..5");
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int BR = BN / RPR;                       // weight staging rounds per step
    static_assert(BR >= 1 && TM >= 1 && TN >= 1 && BN % RPR == 0, "tile shape");
    constexpr int B_BYTES = BN * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    const int tiles_x = (d.W + 15) >> 4, tiles_y = (d.H + PH - 1) / PH;
    const int tpi = tiles_x * tiles_y, ntm = d.N * tpi, ntn = d.CoutPad / BN;
    int tile_m, tile_n;
    if (!gdt_tile_of_block(blockIdx.x, ntm, ntn, tile_m, tile_n)) return;      // XCD-chunked, see gdt_common.h
    const int n = tile_m / tpi, tr = tile_m - n * tpi;
    const int y0 = (tr / tiles_x) * PH, x0 = (tr % tiles_x) << 4;

    // ---- loader state
    // LDS swizzle of the halo image: chunk' = chunk ^ ((halo column >> 1) & 7).  With 18-pixel halo rows this keeps every
    // 16-lane ds_read_b128 group of a fragment read (two patch rows, 16 + 16 pixels) on 16 distinct 16-byte slots for all
    // nine taps (the generic (row >> 1) & 7 swizzle is 2-way conflicted here: SQ_LDS_BANK_CONFLICT was 38 % of LDS cycles).
    const int lrow = tid >> 3;                          // 0..RPR-1
    const int q = (lane & 7) ^ ((lrow >> 1) & 7);       // weight tile: generic swizzle
    const bool refl = d.pad_reflect != 0;
    int a_pix[NR], a_q[NR]; unsigned a_ok = 0, a_int = 0;     // a_int: piece is an interior pixel of this patch (inside the image)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int h = r * RPR + lrow;
        const int hy = h / HALO_W, hx = h - hy * HALO_W;
        a_q[r] = (lane & 7) ^ ((hx >> 1) & 7);
        a_int |= (((hy >= 1) & (hy <= PH) & (hx >= 1) & (hx <= 16) & (y0 - 1 + hy < d.H) & (x0 - 1 + hx < d.W)) ? 1u : 0u) << r;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        int ry = iy < 0 ? -iy : (iy >= d.H ? 2 * d.H - 2 - iy : iy);
        int rx = ix < 0 ? -ix : (ix >= d.W ? 2 * d.W - 2 - ix : ix);
        ry = min(max(ry, 0), d.H - 1); rx = min(max(rx, 0), d.W - 1);
        const bool inb = ((unsigned)iy < (unsigned)d.H) & ((unsigned)ix < (unsigned)d.W);
        a_pix[r] = (n * d.H + ry) * d.W + rx;
        a_ok |= ((h < HALO_ROWS) & (inb | refl) ? 1u : 0u) << r;
    }
    const f16* b_src = d.w + ((long)(tile_n * BN + lrow) * d.Kpad + q * 8);

    auto issue_a = [&](int chunk, int stage, int r) {
        if (r >= NR || r * RPR + wave * 8 >= HALO_ROWS_PAD) return;  // wave-uniform: rows beyond the padded halo
        const f16* src = d.in + (((long)a_pix[r] << (d.lc8 + 3)) + (chunk * 8 + a_q[r]) * 8);
        glds16(((a_ok >> r) & 1u) ? src : d.zeros, smem + stage * A_BYTES + (r * RPR + wave * 8) * ROWB);
    };
    // Fused InstanceNorm (+ReLU) of the producer (p2p_networks.py:29,:272): when d.in_norm is set the halo goes through
    // registers instead -- load 8 raw fp16 channels, x -> max((x - mean) * rstd, 0) in fp32, store to the same swizzled LDS
    // slot the DMA path would have filled.  One piece per tap step, written one step after it was issued.
    const bool norm_a = d.in_norm != nullptr;     // wave-uniform
    // (mean, rstd) of the 64 channels of a chunk are staged once per chunk in LDS (512 B per stage, behind the weight
    // stages) and read per piece from there: the per-piece channel group depends on the halo column swizzle, and fetching it
    // from global memory per piece cost as many L2 bytes as the weight tile itself.
    float* nlds = (float*)(smem + 2 * A_BYTES + 2 * B_BYTES);
    auto stage_norm = [&](int chunk) {          // 32 lanes x float4 = 64 channels x (mean, rstd)
        if (tid < 32) {                         // kept as (scale, shift) = (rstd, -mean * rstd): one fma per element
            const float4 v = *(const float4*)(d.in_norm + ((long)n * d.Cin + chunk * 64) * 2 + tid * 4);
            *(float4*)(nlds + (chunk & 1) * 128 + tid * 4) = make_float4(v.y, -v.x * v.y, v.w, -v.z * v.w);
        }
    };
    auto load_piece = [&](int chunk, int r) -> f16x8 {
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (f16)0.f;
        if (r < NR && r * RPR + lrow < HALO_ROWS_PAD && ((a_ok >> r) & 1u))
            v = *(const f16x8*)(d.in + (((long)a_pix[r] << (d.lc8 + 3)) + (chunk * 8 + a_q[r]) * 8));
        return v;
    };
    auto load_res_piece = [&](int chunk, int r) -> f16x8 {
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (f16)0.f;
        if (d.in_res && r < NR && r * RPR + lrow < HALO_ROWS_PAD && ((a_ok >> r) & 1u))
            v = *(const f16x8*)(d.in_res + (((long)a_pix[r] << (d.lc8 + 3)) + (chunk * 8 + a_q[r]) * 8));
        return v;
    };
    auto store_piece = [&](int stage, int chunk, int r, const f16x8& raw, const f16x8& resv) {
        const int row = r * RPR + lrow;
        if (r >= NR || row >= HALO_ROWS_PAD) return;
        float nmr[16];
        const float4* np4 = (const float4*)(nlds + stage * 128 + a_q[r] * 16);      // stage == chunk & 1
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float4 v = np4[k]; nmr[4 * k] = v.x; nmr[4 * k + 1] = v.y; nmr[4 * k + 2] = v.z; nmr[4 * k + 3] = v.w; }
        f16x8 o;
        const bool ok = (a_ok >> r) & 1u;                      // padded positions stay exactly zero
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = fmaf((float)raw[e], nmr[2 * e], nmr[2 * e + 1]);
            if (d.in_relu) f = fmaxf(f, 0.f);
            if (d.in_res) f += (float)resv[e];
            o[e] = ok ? (f16)f : (f16)0.f;
        }
        *(f16x8*)(smem + stage * A_BYTES + row * ROWB + ((lane & 7) << 4)) = o;
        // the transformed tensor itself (e.g. the ResnetBlock output) is materialised by the patch that owns the pixel
        if (d.in_out && tile_n == 0 && ((a_int >> r) & 1u))
            *(f16x8*)(d.in_out + (((long)a_pix[r] << (d.lc8 + 3)) + (chunk * 8 + a_q[r]) * 8)) = o;
    };
    auto issue_b = [&](int koff, int stage, int r) {
        glds16(b_src + ((long)r * RPR * d.Kpad + koff), smem + 2 * A_BYTES + stage * B_BYTES + (r * RPR + wave * 8) * ROWB);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    // fragment addresses: row base | ((first chunk ^ swizzle) << 4); the k-substep kk is applied with ONE xor (kk << 5),
    // because (2*kk + fh) ^ sw == (2*kk) ^ (fh ^ sw) and the row bases are multiples of 128
    int a_h0[TM], a_px[TM], b_base[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = wm * WTM + i * 32 + fr;
        a_h0[i] = (m >> 4) * HALO_W + (m & 15); a_px[i] = m & 15;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + fr;
        b_base[j] = 2 * A_BYTES + row * ROWB + ((fh ^ ((row >> 1) & 7)) << 4);
    }

    const int nchunks = d.Cin >> 6;
    const int total = nchunks * 9;
    if (norm_a) {
        stage_norm(0);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NR; ++r) store_piece(0, 0, r, load_piece(0, r), load_res_piece(0, r));
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) issue_a(0, 0, r);
    }
#pragma unroll
    for (int r = 0; r < BR; ++r) issue_b(0, 0, r);
    f16x8 pend, pend_res;
#pragma unroll
    for (int e = 0; e < 8; ++e) { pend[e] = (f16)0.f; pend_res[e] = (f16)0.f; }

    f16x8 afr[2][TM], bfr[2][TN];
    int c = 0, t = 0;                                   // chunk, tap of the current step
    // diagnostic build only (GDT_CONV_STAMP=1 -> d.dbg & 16): cycles spent waiting at the step barrier vs in the step body
    unsigned long long st_bar = 0, st_body = 0, st_prev = 0;
    const bool stamp = (d.dbg & 16) != 0;
    if (stamp) st_prev = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < total; ++s) {
        unsigned long long st0 = 0;
        if (stamp) { st0 = __builtin_amdgcn_s_memtime(); st_body += st0 - st_prev; }
        __syncthreads();
        if (stamp) { st_prev = __builtin_amdgcn_s_memtime(); st_bar += st_prev - st0; }
        const bool more = (s + 1 < total) && !(d.dbg & 1);
        int nc = c, nt = t + 1;
        if (nt == 9) { nt = 0; nc = c + 1; }
        const int nkoff = nt * d.Cin + (nc << 6);         // K offset of the next step's weight slice
        const bool halo_more = (c + 1 < nchunks) && t < NR && !norm_a;
        if (norm_a && c + 1 < nchunks && t == 0) stage_norm(c + 1);           // visible after the next barrier
        const int ty = (t * 21846) >> 16, tx = t - ty * 3;
        int a_ad[TM], b_ad[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            a_ad[i] = (c & 1) * A_BYTES + (a_h0[i] + ty * HALO_W + tx) * ROWB + ((fh ^ (((a_px[i] + tx) >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) b_ad[j] = b_base[j] + (s & 1) * B_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) afr[0][i] = *(const f16x8*)(smem + a_ad[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[0][j] = *(const f16x8*)(smem + b_ad[j]);
        if (d.dbg & 2) {       // timing-only ablation: staging without MFMAs
            if (halo_more) issue_a(c + 1, (c + 1) & 1, t);
            if (more) {
#pragma unroll
                for (int r = 0; r < BR; ++r) issue_b(nkoff, (s + 1) & 1, r);
            }
            c = nc; t = nt;
            continue;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i) afr[nxt][i] = *(const f16x8*)(smem + (a_ad[i] ^ ((kk + 1) << 5)));
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[nxt][j] = *(const f16x8*)(smem + (b_ad[j] ^ ((kk + 1) << 5)));
            }
            if (kk == 1 && halo_more) issue_a(c + 1, (c + 1) & 1, t);
            // normalised halo: the VALU work of a piece sits in the middle of the step's MFMA stream (right after the step
            // barrier every wave would do it at once and the matrix pipe would idle); the piece was loaded one step earlier
            if (kk == 2 && norm_a && c + 1 < nchunks) {
                if (t >= 1 && t <= NR) store_piece((c + 1) & 1, c + 1, t - 1, pend, pend_res);
                if (t < NR) { pend = load_piece(c + 1, t); pend_res = load_res_piece(c + 1, t); }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
                    // next step's weight tile: one staging round after every second MFMA from the start of the step, so the
                    // DMA has most of the step to land before the barrier that publishes it
                    {
                        constexpr int PER_KK = (TM * TN + 1) / 2;                 // issue slots per k-substep
                        const int slot = kk * PER_KK + (i * TN + j) / 2;
                        const bool at_slot = (TM * TN == 1) ? true : ((i * TN + j) % 2 == 1);
                        if (more && at_slot && slot < BR) issue_b(nkoff, (s + 1) & 1, slot);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        c = nc; t = nt;
    }

    // ---------------------------------------------------------------- epilogue (fp16 NHWC through an LDS transpose)
    unsigned long long st_loop_end = 0;
    if (stamp) { st_loop_end = __builtin_amdgcn_s_memtime(); st_body += st_loop_end - st_prev; }
    if (d.dbg & 4) return;            // timing-only ablation
    conv_epilogue_f16<BM, BN, WGM, WGN, NT, TM, TN>(d, acc, smem, tile_m, tile_n, [&](int row, bool& ok) -> long {
        const int y = y0 + (row >> 4), x = x0 + (row & 15);
        ok = (y < d.H) & (x < d.W);
        return ((long)n * d.H + y) * d.W + x;
    });
    if (stamp && lane == 0 && d.stamp_out) {
        const unsigned long long st_end = __builtin_amdgcn_s_memtime();
        unsigned long long* o = d.stamp_out + ((long)blockIdx.x * 8 + wave) * 4;
        o[0] = st_bar; o[1] = st_body; o[2] = st_end - st_loop_end; o[3] = total;
    }
}

template <int PH, int BN, int WGM, int WGN>
int launch_halo(const ConvLaunch& d, hipStream_t stream) {
    const int tiles = d.N * ((d.W + 15) / 16) * ((d.H + PH - 1) / PH), ntn = d.CoutPad / BN;
    constexpr size_t lds = halo_lds_bytes<PH, BN, WGM, WGN>();
    static_assert(lds <= 160 * 1024, "LDS budget");
    static GdtPerDevice per_dev;          // one attribute call per template instantiation AND device (gdt_common.h)
    int attr_set = 0;
    {
        const int rc = gdt_per_device(per_dev, attr_set, [](int, int, int& v) {
            v = 1;
            GDT_CHECK_HIP(hipFuncSetAttribute((const void*)conv3x3_halo_kernel<PH, BN, WGM, WGN>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            return GDT_OK;
        });
        if (rc != GDT_OK) return rc;
    }
    hipLaunchKernelGGL((conv3x3_halo_kernel<PH, BN, WGM, WGN>), dim3(gdt_grid_for_tiles(tiles, ntn)), dim3(WGM * WGN * 64), lds, stream, d);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // namespace

// Eligibility: 3x3, stride 1, pad 1, Cin a multiple of 64, fp16 NHWC output, enough tiles to fill the chip, and -- when the
// InstanceNorm statistics are fused -- whole 16x16 patches (H, W multiples of 16) so that the 128-row records line up.
bool gdt_conv_halo_eligible(const ConvLaunch& d) {
    static const int mode = [] { const char* e = getenv("GDT_CONV_HALO"); return e ? atoi(e) : 1; }();   // 0 off, 1 auto, 2 force
    if (mode == 0) return false;
    const bool shape = d.ntaps == 9 && d.TW == 3 && d.sy == 1 && d.sx == 1 && d.dy0 == -1 && d.dx0 == -1 && d.dys == 1 && d.dxs == 1 &&
                       d.osy == 1 && d.osx == 1 && d.ooy == 0 && d.oox == 0 && d.Cin % 64 == 0 && !d.out_f32 && d.Cout % 8 == 0 &&
                       d.OH == d.H && d.OW == d.W && d.Kpad == 9 * d.Cin && d.CoutPad % 64 == 0;
    if (!shape) return false;
    if (d.stats && ((d.H & 15) || (d.W & 15))) return false;     // whole patches: the 128-row statistics records line up
    if (mode == 2) return true;
    const long tiles = (long)d.N * ((d.W + 15) / 16) * ((d.H + 15) / 16);
    const int bn = d.CoutPad % 256 == 0 ? 256 : (d.CoutPad % 128 == 0 ? 128 : 64);
    // padded patches waste work on ragged sizes: require >= 85 % useful pixels
    const double useful = (double)d.H * d.W / ((double)((d.H + 15) / 16 * 16) * ((d.W + 15) / 16 * 16));
    static const int min_env = [] { const char* e = getenv("GDT_CONV_MIN_TILES"); return e ? atoi(e) : 0; }();
    // (batch-1 sweeps of the first-generation kernel: 64-128 best, 512 loses 25 % on a 1024^2 image.  With fragment-ordered weights the layer runs on
    // conv3x3_halo_rb.hip, whose 128-column four-wave tiles keep winning down to 16 patches: GeM-ResNet-101 8 x 512^2 2.38 -> 2.15 ms, 1 x 1024^2 2.17 -> 2.03)
    const int min_tiles = min_env ? min_env : ((d.w_frag && !d.in_norm && !d.stats) ? 16 : 128);       // (the folded-norm modes only exist in the eight-wave form)
    return tiles * (d.CoutPad / bn) >= min_tiles && useful >= 0.85;
}

int gdt_launch_conv_halo(const ConvLaunch& d_in, hipStream_t stream) {
    static const int dbg = [] { const char* e = getenv("GDT_CONV_DBG"); return e ? atoi(e) : 0; }();
    ConvLaunch d = d_in;
    d.dbg = dbg;
    static const int want_stamp = [] { const char* e = getenv("GDT_CONV_STAMP"); return e ? atoi(e) : 0; }();
    static unsigned long long* stamp_buf = nullptr;
    const int stamp_blocks = (d.N * ((d.W + 15) / 16) * ((d.H + 15) / 16) + 7) / 8 * 8 * (d.CoutPad / 256 ? d.CoutPad / 256 : 1);
    if (want_stamp && d.CoutPad % 256 == 0) {            // diagnostic: per-wave cycle totals printed after a blocking sync
        if (!stamp_buf) GDT_CHECK_HIP(hipMalloc((void**)&stamp_buf, (size_t)65536 * 8 * 4 * sizeof(unsigned long long)));
        GDT_CHECK_HIP(hipMemsetAsync(stamp_buf, 0, (size_t)stamp_blocks * 8 * 4 * sizeof(unsigned long long), stream));
        d.dbg |= 16; d.stamp_out = stamp_buf;
        const int rc = launch_halo<16, 256, 2, 4>(d, stream);
        GDT_CHECK_HIP(hipStreamSynchronize(stream));
        std::vector<unsigned long long> h((size_t)stamp_blocks * 8 * 4);
        GDT_CHECK_HIP(hipMemcpy(h.data(), stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        double bar = 0, body = 0, epi = 0, nw = 0, steps = 0;
        for (size_t w = 0; w < h.size() / 4; ++w) if (h[w * 4 + 3]) { bar += h[w * 4]; body += h[w * 4 + 1]; epi += h[w * 4 + 2]; steps += h[w * 4 + 3]; nw += 1; }
        fprintf(stderr, "[halo stamp] waves %.0f: per step: barrier %.0f cyc, body %.0f cyc; epilogue %.0f cyc per wave; steps/wave %.0f\n", nw,
                bar / steps, body / steps, epi / nw, steps / nw);
        return rc;
    }
    static const int small = [] { const char* e = getenv("GDT_HALO_SMALL"); return e ? atoi(e) : 0; }();   // experiment knob
    if (small == 1 && d.CoutPad % 128 == 0) return launch_halo<8, 128, 2, 2>(d, stream);  // two workgroups per CU
    // 16 wavefronts (64x64 per wave, 4 per SIMD): equal to the 8-wave form on plain layers (0.281 vs 0.283 ms).  With the
    // producer's InstanceNorm applied while staging it is the slower one (0.51 vs 0.32 ms) since the staging moved into the
    // MFMA stream: at 128 registers per lane the extra live values spill.  Kept as an experiment knob.
    if (small == 2 && d.CoutPad % 256 == 0) return launch_halo<16, 256, 4, 4>(d, stream);
    if (d.CoutPad % 256 == 0) return launch_halo<16, 256, 2, 4>(d, stream);
    if (d.CoutPad % 128 == 0) return launch_halo<16, 128, 4, 2>(d, stream);
    return launch_halo<16, 64, 8, 1>(d, stream);
}
