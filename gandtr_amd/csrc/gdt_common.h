// Shared declarations for the gandtr HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <mutex>
#include <string>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (thread-local message, integer status; see include/gandtr_hip.h) ----
void gdt_set_error(const std::string& msg);
#define GDT_OK 0
#define GDT_ERR_INVALID 1
#define GDT_ERR_HIP 2
#define GDT_ERR_WORKSPACE 3
#ifndef GDT_ERR_NOT_CONVERGED
#define GDT_ERR_NOT_CONVERGED 4
#endif

#define GDT_CHECK_HIP(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            gdt_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                   \
            return GDT_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

#define GDT_REQUIRE(cond, msg)                                                                  \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            gdt_set_error(std::string("invalid argument: ") + msg + " [" #cond "]");            \
            return GDT_ERR_INVALID;                                                             \
        }                                                                                       \
    } while (0)

// ---- one-time set-up of a kernel PER DEVICE ----
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current device only, and a process may build nets on several GPUs and launch from
// several threads (the JPEG staging explicitly supports that): the launchers keep their "done" state per device id behind a mutex.  `setup(dev, cus,
// value)` runs once per device with the device's CU count and leaves the launcher's cached figure in `value` (non-zero).
struct GdtPerDevice {
    std::mutex mu;
    int value[64] = {};
};
template <typename Setup>
inline int gdt_per_device(GdtPerDevice& st, int& out, Setup&& setup) {
    int dev = 0;
    GDT_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) {
        gdt_set_error("device id out of range");
        return GDT_ERR_INVALID;
    }
    std::lock_guard<std::mutex> lock(st.mu);
    if (!st.value[dev]) {
        int cus = 0, v = 0;
        GDT_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int rc = setup(dev, cus, v);
        if (rc != GDT_OK) return rc;
        st.value[dev] = v ? v : 1;
    }
    out = st.value[dev];
    return GDT_OK;
}

// ---- implicit-GEMM convolution launch descriptor ----
// Activations are NHWC fp16; weights are packed [CoutPad][Kpad] fp16 with k = tap * Cin + c.
// A "tap" t has input offset (dy, dx) = (dy0 + (t / TW) * dys, dx0 + (t % TW) * dxs); an output-grid position
// (oy, ox) reads input pixel (oy * sy + dy, ox * sx + dx) and writes output pixel (oy * osy + ooy, ox * osx + oox).
// That covers Conv2d (any k, stride 1/2, zero or reflect padding) and the four sub-pixel phases of
// ConvTranspose2d(k3, s2, p1, op1).
struct ConvLaunch {
    const f16* in;        // [N][H][W][Cin]
    const f16* w;         // [CoutPad][Kpad]
    const f16* w_lo;      // f16x3 mode: low parts of the weights, (w - fp16(w)) * 2^11, same layout; else nullptr
    const float* bias;    // [CoutPad] or nullptr
    const f16* res;       // residual, same layout as out, or nullptr
    f16* out;             // [N][OH][OW][Cout]  (nullptr when out_f32 is used)
    float* out_f32;       // [N][Cout][OH][OW] fp32 NCHW, or nullptr
    float* stats;         // per-tile partial sums [tiles][2][Cout] (sum, sum of squares) for InstanceNorm, or nullptr
    const f16* zeros;     // >= 16 B of zeros (source for padded / out-of-range chunks)
    const float* in_norm; // optional fused InstanceNorm of the INPUT: (mean, rstd) pairs [N][Cin][2], applied while staging A
    int in_relu;          // ... followed by ReLU
    const f16* w_frag;    // weights in MFMA B-fragment order [CoutPad/32][Kpad/16][64 lanes][8] (3x3 s1 p1 layers, else null)
    const f16* in_res;    // ... then + residual (ResnetBlock output y = x + IN(conv), p2p_networks.py:505), same layout as `in`
    f16* in_out;          // ... and the transformed input is ALSO written here (each patch writes its interior pixels)
    int N, H, W, Cin, lc8;        // lc8 = log2(Cin / 8)
    int Cout, CoutPad, Kpad, nk;  // nk = Kpad / 64
    int OHg, OWg, OH, OW;         // output grid of this launch, full output size
    int sy, sx, osy, ooy, osx, oox;
    int ntaps, TW, invTW, dy0, dys, dx0, dxs;
    int pad_reflect;              // 0: zero padding, 1: reflect
    int relu;                     // fused ReLU (after bias and residual)
    int act;                      // out_f32 only: 0 none, 1 tanh, 2 sigmoid
    int M;                        // N * OHg * OWg
    int dbg;                      // timing-only ablation knob (env GDT_CONV_DBG): 1 skip staging loads, 2 skip MFMAs
    unsigned long long* stamp_out;  // diagnostic builds only (GDT_CONV_STAMP): per-wave s_memtime totals
    int stats_tile_base;          // tile index offset for this launch in the stats slab (ConvTranspose phases)
    int pool2;                    // 1: MaxPool2d(2, 2) fused into the epilogue of the patch kernels; `out` is the pooled [N][H/2][W/2][Cout] tensor
    // "f16c" precision mode (conv3x3_halo_c.hip): block-scaled correction operands of the weights in MFMA fragment order --
    // wmx_a [CoutPad/128][Kpad/32][4][64 lanes][16 B] + wmx_b [..][64][8 B]: per lane 32 e2m3 values (24 bytes); wmx_s [..][64] dwords: its E8M0
    // block scale (lanes 0-31: fp16(w) of output channel lane, lanes 32-63: w - fp16(w), same 32 k-values).  Activation side: a_lo is stored as fp4(a_lo * 2^c_lo_exp), a_hi as fp4(a_hi * 2^-c_hi_exp).
    const f16* w_frag2;           // conv_stem.hip, f16c form: the weight residuals W2 = [w - fp16(w), 0 ..] in the stem fragment order
    const void* w_cfrag;          // fp16 weights grouped per 128 output channels: [CoutPad/128][Kpad/16][4][64 lanes][8 halves] (wmx_* grouped alike)
    const void* wmx_a; const void* wmx_b; const void* wmx_s;
    // conv3x3_halo_c16.hip (the same layer on the 16 x 16 MFMA shapes): ONE 15 KB record per (64 output channels, 64 k-values), index
    // (cout / 64) * (Kpad / 64) + k / 64: [fp16 weights of k 0-31: 4 blocks of 16 channels x 64 lanes x 16 B, lane (n, g) = channel block * 16 + n,
    // k = 8 g ..+7][the same of k 32-63][correction operands, first 16 bytes per lane: 4 blocks x 64 lanes][their last 8 bytes][E8M0 scales: 64 lanes x
    // 4 blocks]; lane (n, blk) of a correction operand = 32 e2m3 values of fp16(w) (blk 0, 2) / w - fp16(w) (blk 1, 3) of k 0-31 (blk 0, 1) / 32-63 (blk 2, 3)
    const void* w_c16;
    int c_lo_exp, c_hi_exp;
    int pair_cout, ooy2, oox2;    // conv3x3_halo_x3.hip FORM 1: > 0 = the 128 GEMM columns are TWO sub-pixel phases of a transposed conv with pair_cout (64) output channels each;
                                  //      the second half writes output pixel (oy * osy + ooy2, ox * osx + oox2) and its statistics one record set (M / 128 records) further on
    int x3_form;                  // conv3x3_halo_x3.hip: 2 = Conv2d(k3,s2,p1) as 2 x 2 shifts over the virtual space-to-depth view (Cin counts the 4 parities); else 0
    int in_f32;                   // conv_head7.hip: `in` is the fp32 NHWC tensor of the f16c mode (rounded to fp16 once, while staging)
    int stagger_us;               // conv3x3_halo_c.hip: start-up delay step between the four workgroup phase groups (0: none)
    int phase_cout;               // > 0: fused ConvTranspose2d(k3,s2,p1,op1) -- GEMM column = phase * phase_cout + cout, phase = py * 2 + px,
                                  //      written to output pixel (2y + py, 2x + px); Cout / CoutPad count GEMM columns (conv_igemm_rb.hip)
    // conv1x1_rb.hip, K-concatenated form: a second 1x1 conv (stride in2_stride, its own input tensor in2 [N][in2_h][in2_w][in2_cin]) accumulated into
    // the same tile -- out = W . in + W2 . in2 with the weights packed as ONE [CoutPad][Cin + in2_cin] matrix (Kpad = Cin + in2_cin): the projection
    // shortcut of a ResNet Bottleneck folded into its expand conv
    const f16* in2; int in2_cin, in2_h, in2_w, in2_stride;
    // conv3x3_expand_rb.hip: the Bottleneck's expand conv run on the 3x3's tile while it is in LDS -- out = ReLU(res + x_w . ReLU(conv3x3 + bias) + x_bias);
    // x_w_frag in the fragment order of w_frag with K = Cout of the 3x3; x_bias: the expand's bias as one weight fragment per 32 channels (lane (c, 0) =
    // { fp16(b), fp16(b - fp16(b)), 0 .. }: it is added by an MFMA against a { 1, 1, 0 .. } pixel operand); `out`, `res` have x_cout channels
    const f16* x_w_frag; const f16* x_bias; int x_cout;
    // ... CHAIN form: the NEXT block's reduce conv (1x1, x_cout -> 256, bias, ReLU) on the tile just written: r_w_frag in the fragment order of w_frag with K = x_cout,
    // r_bias [256] fp32, r_out [N][H][W][256]
    const f16* r_w_frag; const float* r_bias; f16* r_out;
    // planner hint: this geometry runs CONCURRENTLY with others of the same net (the levels of a pyramid on side streams): (pixels of all of them) / (its own), >= 1.
    // Fusions whose tile thresholds say "enough patches to fill the chip" count the group's patches (conv3x3_expand_rb.hip)
    float group_factor;
};

// ---- several independent geometries in ONE launch (round 5: the levels of a multi-scale pyramid, wrapper.py:221-233 / network.py:139-140) ----
// Every level keeps its own launch descriptor (its own pointers, N, H, W: the levels need not be contiguous); the workgroups of the grid are dealt out to the
// levels in contiguous ranges proportional to their tiles (whole multiples of 8, so that the XCD of a workgroup -- blockIdx & 7 -- is what its level's tile walk
// assumes) and every workgroup runs the unchanged kernel body on its level's descriptor with its index and its level's grid size in place of blockIdx /
// gridDim.  Results are those of the levels launched one by one with the same kernel instantiation, bit for bit.
// MEASURED (GeM-ResNet-101, 8 x 1024^2, hub scales; tools/ab_levels.py, alternating inside one process): 82 of the 88 launches the three levels hand in are
// joined, 120 launches instead of 300 -- and the forward takes 7.44 ms against 7.09 ms with one side stream per level (sms preset 15.9 vs 14.0): the 1x1 convs
// have thousands of tiles per level, so a joined launch takes the SUM of its levels' times like the streams do, a level has its share of the chip for the
// whole op instead of what is free at the moment, and the ranges' rounding to XCD rounds adds a tail per op.  The other arrangement -- every workgroup walks
// the levels' tile lists one after the other with the whole grid -- measured 10.1 ms (the level-by-level sum).  The streams stay the default
// (GANDTR_HIP_JOINT_LEVELS=1 selects this path).
constexpr int GDT_MAX_LEVELS = 4;
struct MultiConv {
    int nlev;
    int prefix[GDT_MAX_LEVELS + 1];      // first workgroup of each level; prefix[nlev] = grid size
    int vblocks[GDT_MAX_LEVELS];         // virtual blocks (tile walk length) of each level
    ConvLaunch lev[GDT_MAX_LEVELS];
};
static_assert(sizeof(MultiConv) <= 4096, "kernel arguments are limited to 4 KB");
__device__ __forceinline__ int gdt_multi_level(const int nlev, const int* prefix, const int b) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < GDT_MAX_LEVELS; ++k) l += (k < nlev && b >= prefix[k]) ? 1 : 0;
    return l;
}
// host: grid ranges per level -- every level all its virtual blocks when they fit `slots` together, else shares in proportion (multiples of 8, at least 8)
inline int gdt_multi_partition(int* prefix, const int* vb, int L, int slots) {
    long total = 0;
    for (int l = 0; l < L; ++l) total += vb[l];
    prefix[0] = 0;
    for (int l = 0; l < L; ++l) {
        int g = vb[l];
        if (total > slots) {
            g = (int)((long)slots * vb[l] / total) / 8 * 8;
            if (g < 8) g = 8;
            if (g > vb[l]) g = vb[l];
        }
        prefix[l + 1] = prefix[l] + g;
    }
    return prefix[L];
}

// variant (optional out): which kernel ran -- BM*1000+BN for conv_igemm_kernel<BM,BN,..>, 900000+BN for conv3x3_halo_kernel<BN,..>, 910000+BN for conv3x3_halo_rb_kernel<BN,..>
// Workgroup -> (M tile, N tile).  Workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md), each
// with a private L2.  Every XCD therefore gets ONE contiguous span of M tiles: spatially adjacent tiles (which share input
// rows through the kernel taps / halos) and the N tiles of one M tile (which share the whole A operand) meet in the same L2.
// With tiles interleaved over the XCDs instead, the 7x1 head conv of the generator fetched its 537 MB input 6.7 times
// (FETCH_SIZE 3.6 GB per launch).  Placement only affects speed, never results.  Grid size: 8 * ceil(ntm / 8) * ntn.
__device__ __forceinline__ bool gdt_tile_of_block(int b, int ntm, int ntn, int& tile_m, int& tile_n) {
    const int mchunk = (ntm + 7) >> 3;
    const int xcd = b & 7, j = b >> 3;
    tile_n = j % ntn;
    const int lm = j / ntn;
    tile_m = xcd * mchunk + lm;
    return lm < mchunk && tile_m < ntm;
}
inline int gdt_grid_for_tiles(int ntm, int ntn) { return 8 * ((ntm + 7) / 8) * ntn; }

int gdt_launch_conv(const ConvLaunch& d, hipStream_t stream, int* variant = nullptr);
int gdt_conv_family(const ConvLaunch& d);                  // conv_igemm.hip: 1 conv1x1_rb, 2 conv3x3_halo_rb, 0 other (the families with a multi-geometry entry)
bool gdt_conv_igemm_norm_eligible(const ConvLaunch& d);    // conv_igemm.hip: fused input InstanceNorm in the generic kernel
bool gdt_conv_halo_eligible(const ConvLaunch& d);          // conv3x3_halo.hip
int gdt_launch_conv_halo(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_halo_rb_eligible(const ConvLaunch& d);       // conv3x3_halo_rb.hip (weights streamed into registers)
bool gdt_conv3x3_expand_eligible(const ConvLaunch& d);     // conv3x3_expand_rb.hip (3x3 + expand 1x1 + residual of a Bottleneck, variant 939000 + x_cout / 8)
int gdt_launch_conv3x3_expand(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv3x3_expand_chain_eligible(const ConvLaunch& d);   // ... with the next block's reduce conv as a third phase of the same launch (variant 938000 + x_cout / 8)
int gdt_launch_conv_halo_rb(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_halo_rb_levels_ok(const ConvLaunch* dl, int L);                          // ... the plain form on L independent geometries in one launch (MultiConv)
int gdt_launch_conv_halo_rb_levels(const ConvLaunch* dl, int L, hipStream_t stream);
bool gdt_conv_pool2_eligible(const ConvLaunch& d);         // conv_igemm.hip: can this launch (pool2 = 0) take a fused 2x2 max pool?
bool gdt_conv_halo_ct_eligible(const ConvLaunch& d);       // conv3x3_halo_rb.hip, transposed form (variant 960256)
int gdt_launch_conv_halo_ct(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_igemm_rb_eligible(const ConvLaunch& d);      // conv_igemm_rb.hip (persistent implicit GEMM, variant 940000 + BN)
int gdt_launch_conv_igemm_rb(const ConvLaunch& d, hipStream_t stream, int* variant);
int gdt_conv_igemm_rb_stats_sets(const ConvLaunch& d);
bool gdt_conv_stem_pair_eligible(const ConvLaunch& d);    // conv_stem.hip, ResNet stem straight from the fp32 NCHW image (variant 951049)
int gdt_launch_conv_stem_pair(const ConvLaunch& d, const float* x, int C, const int* perm, const float* scale, const float* shift, hipStream_t stream);
int gdt_launch_conv_stem_pair_pool(const ConvLaunch& d, const float* x, int C, const int* perm, const float* scale, const float* shift, int PH, int PW, hipStream_t stream);   // ... + MaxPool2d(3, 2, 1) (variant 952049)
bool gdt_conv_1x1_rb_eligible(const ConvLaunch& d);        // conv1x1_rb.hip (streaming 1x1 conv, variant 945128)
int gdt_launch_conv_1x1_rb(const ConvLaunch& d, hipStream_t stream);
int gdt_launch_conv_1x1_rb_levels(const ConvLaunch* dl, int L, hipStream_t stream);     // ... on L independent geometries in one launch (MultiConv)
bool gdt_conv_1x1_cat_eligible(const ConvLaunch& d);       // ... its K-concatenated form (variant 946128)
// fused transposed conv (phase_cout > 0): GEMM column c -> (sub-pixel phase, output channel).  Each 64-column wave slice pairs
// a cheap phase with an expensive one -- 32 columns of phase 0 (1 input shift) + 32 of phase 3 (4 shifts), or 1 + 2 (2 + 2) --
// so that skipping the all-zero (shift, phase) weight blocks leaves every wave 4-5 of its 8 block-steps.
inline __host__ __device__ void gdt_ctf_column(int c, int phase_cout, int& phase, int& co) {
    const int wq = c >> 6, j = (c >> 5) & 1, wpp = phase_cout >> 5;      // wave slice index, block in slice, slices per pair
    const int pair = (wq / wpp) & 1;
    phase = pair == 0 ? (j ? 3 : 0) : (j ? 2 : 1);
    co = (wq % wpp) * 32 + (c & 31);
}
// ... in conv3x3_halo_c.hip (f16c mode): a wave's 128-column slice holds the four phases of 32 output channels, block j = phase j
inline __host__ __device__ void gdt_ctc_column(int c, int& phase, int& co) {
    phase = (c >> 5) & 3;
    co = (c >> 7) * 32 + (c & 31);
}
bool gdt_conv_stem_eligible(const ConvLaunch& d);          // conv_stem.hip (image -> 64 channels, variant 950000 + taps)
int gdt_launch_conv_stem(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_stem_c_eligible(const ConvLaunch& d);        // ... f16c form: augmented pixel words + residual weights, fp32 output (variant 955000 + taps)
int gdt_launch_conv_stem_c(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_head7_eligible(const ConvLaunch& d);         // conv_head7.hip (fused 7x7 generator head, variant 920007)
int gdt_launch_conv_head7(const ConvLaunch& d, hipStream_t stream);
// f16x3 precision mode (conv_igemm_x3.hip): in / res / out are fp32 NHWC (passed through the f16* fields), nk = Kpad / 32
int gdt_launch_conv_x3(const ConvLaunch& d, hipStream_t stream, int* variant = nullptr);
bool gdt_conv_halo_x3_eligible(const ConvLaunch& d);       // conv3x3_halo_x3.hip
bool gdt_conv_halo_x3_taps_eligible(const ConvLaunch& d);  // ... FORM 1 / 2: tap tables inside the 3 x 3 window (transposed-conv phases), stride 2 over the space-to-depth view
int gdt_launch_conv_halo_x3_taps(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_x3_norm_eligible(const ConvLaunch& d);       // conv_igemm_x3.hip: the generic f16x3 GEMM applies the producer's InstanceNorm (+ReLU) while staging
int gdt_launch_conv_halo_x3(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_halo_c_eligible(const ConvLaunch& d);         // conv3x3_halo_c.hip (f16c mode, variant 970256)
int gdt_launch_conv_halo_c(const ConvLaunch& d, hipStream_t stream);
int gdt_conv_halo_c_columns(const ConvLaunch& d);           // output-channel columns per tile of the form that launch picks (256, or 128 for few patches)
bool gdt_conv_halo_c16_eligible(const ConvLaunch& d);       // conv3x3_halo_c16.hip (f16c mode on the 16 x 16 MFMA shapes, variant 971256)
int gdt_launch_conv_halo_c16(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_halo_c_ct_eligible(const ConvLaunch& d);      // ... transposed form (variant 980256)
int gdt_launch_conv_halo_c_ct(const ConvLaunch& d, hipStream_t stream);
bool gdt_conv_halo_c_s2_eligible(const ConvLaunch& d);      // ... stride-2 form over the virtual space-to-depth input (variant 990256)
int gdt_launch_conv_halo_c_s2(const ConvLaunch& d, hipStream_t stream);
bool gdt_bneck_eligible(int cin, int C, int mid, int N, int H, int W);   // conv_bneck.hip: Bottleneck (1x1 -> 3x3 -> 1x1 + shortcut) as one launch (variant 935000 + C)
int gdt_launch_bneck(const f16* x, f16* y, const f16* wr, const f16* w3, const f16* we, const float* br, const float* b3, const float* be,
                     const f16* wd, const float* bd, int cin, int C, int mid, int N, int H, int W, hipStream_t stream);   // wd / bd: 1x1 projection shortcut (cin != C), else null
int gdt_launch_bneck_levels(const f16* const* x, f16* const* y, const f16* wr, const f16* w3, const f16* we, const float* br, const float* b3, const float* be,
                            const f16* wd, const float* bd, int cin, int C, int mid, const int* N, const int* H, const int* W, int L, hipStream_t stream);   // ... on L geometries in one launch
int gdt_conv_bn(int Cout);    // N tile used for a given Cout (CoutPad must be a multiple of it)
