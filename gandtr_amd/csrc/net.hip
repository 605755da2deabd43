// Graph builder + executor behind the C ABI in include/gandtr_hip.h.
//
// The reference executes its models as nn.Sequential / nn.Module graphs (p2p_networks.py:313, imageretrievalnet.py:93,
// hed.py:30-45).  Here the host mirror (gandtr_amd/, Python) describes the same layer graph once through gdt_net_*;
// this file packs the weights (BatchNorm folded, fp16, [CoutPad][taps*Cin]), infers shapes per call, plans a
// liveness-based workspace layout and launches the HIP kernels on the caller's stream.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/gandtr_hip.h"
#include "aux_kernels.h"
#include "gdt_common.h"

static thread_local std::string g_last_error;
void gdt_set_error(const std::string& msg) { g_last_error = msg; }

namespace {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }
inline int next_pow2(int v) { int p = 8; while (p < v) p <<= 1; return p; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

enum OpKind { OP_INPUT, OP_CONV, OP_INORM, OP_MAXPOOL, OP_GEM, OP_OUT_NCHW, OP_HED };

struct PackedPhase {
    size_t w_off = 0;                 // byte offset in the device weight blob
    size_t w_lo_off = 0;              // f16x3 mode: offset of the low parts
    size_t w_frag_off = 0; bool has_frag = false;   // fp16 mode, 3x3 s1 p1: copy in MFMA B-fragment order (conv3x3_halo_rb.hip)
    size_t w_frag2_off = 0; bool has_aug = false;   // f16c stem: w_frag = augmented W1, w_frag2 = residual W2 (conv_stem.hip)
    size_t w_pair_off = 0; bool has_pair = false;   // fp16 ResNet stem (7x7 s2): pair-word k order of conv_stem_pair_kernel
    size_t wc_off = 0, wmx_a_off = 0, wmx_b_off = 0, wmx_s_off = 0; bool has_mx = false;   // f16c mode: block-scaled correction operands (ConvLaunch::wmx_*)
    size_t w16_off = 0; bool has_mx16 = false;   // ... in the 16 x 16 fragment order, one record per (64 channels, 64 k) (ConvLaunch::w_c16)
    int ntaps = 0, TW = 1, dy0 = 0, dys = 1, dx0 = 0, dxs = 1, Kpad = 0;
    int ooy = 0, oox = 0;
    int ooy2 = 0, oox2 = 0;           // paired phases (Op::pairs): output pixel offset of the second half
};

struct Op {
    OpKind kind;
    int in = -1, res = -1, out = -1, slot = -1;
    // conv
    gdt_conv_desc cd{};
    int cin_pad = 0, cout_pad = 0;
    std::vector<PackedPhase> phases;
    size_t bias_off = 0; bool has_bias = false;
    size_t bias_frag_off = 0; bool has_bias_frag = false;    // 1x1 convs from 256 channels: the bias as an MFMA weight fragment (conv3x3_expand_rb.hip)
    // input
    int in_c = 0; int perm[8]; float scale[8], shift[8];
    // inorm
    float eps = 1e-5f; int relu = 0;
    int stats_from = -1;   // INORM: index of the conv op whose epilogue can deliver the statistics
    int stats_for = -1;    // CONV: index of the INORM op consuming this conv's output
    // CONV with few output channels written as fp32 NCHW (generator head): k x 1 implicit GEMM with kw*cout channels into a
    // scratch tensor + horizontal combine (rowsplit_combine_kernel)
    bool rowsplit = false; int rs_cout8 = 0; size_t rs_bias_off = 0;
    // ConvTranspose2d(k3,s2,p1,op1) as ONE GEMM: columns = 4 sub-pixel phases x cout, K = 4 input shifts x cin (conv_igemm_rb.hip)
    bool has_ctf = false; PackedPhase ctf; size_t ctf_bias_off = 0;
    // f16c: Conv2d(k3, s2, p1) as a 2x2-shift conv over the virtual space-to-depth input (conv3x3_halo_c.hip, FORM 2): K = 4 shifts x 4 cin
    bool has_s2 = false; PackedPhase s2; size_t s2_bias_off = 0; int s2_cout_pad = 0;
    // f16x3, ConvTranspose2d(k3,s2,p1,op1) with 64 output channels: the phases (py, 0) and (py, 1) as ONE 128-column GEMM per py over the union of their taps
    // (conv3x3_halo_x3.hip FORM 1 with ConvLaunch::pair_cout): 3/4 of the products are useful, and the input patch is staged twice instead of four times
    bool has_pairs = false; std::vector<PackedPhase> pairs; size_t pair_bias_off = 0;
    // fp16 mode: a 1x1 expand conv whose residual is the output of a 1x1 projection conv (ResNet Bottleneck shortcut, stride 1 or 2) carries the two
    // weight matrices K-concatenated (conv1x1_rb.hip, CAT form): kcat_ds = index of the projection op
    int kcat_ds = -1; size_t kcat_frag_off = 0, kcat_bias_off = 0;
    // maxpool
    int k = 0, s = 0, p = 0;
    // gem
    float gem_p = 3.f, eps_gem = 1e-6f, eps_l2 = 1e-6f;
    // out_nchw
    size_t tap_bias_off = 0; bool tap_has_bias = false;
    // hed
    int feats[5]; size_t score_w_off[5]; float score_b[5], fusion_w[5], fusion_b = 0.f; int sigmoid = 1;
};

struct Tensor { int C = 0; int Creal = 0; int H = 0, W = 0; int last_use = -1; size_t off = 0, bytes = 0; };   // C: padded, Creal: logical

// first-fit allocator with coalescing free list; "top" grows when nothing fits
struct Arena {
    struct Blk { size_t off, size; };
    std::vector<Blk> free_;
    size_t top = 0, peak = 0;
    size_t alloc(size_t bytes) {
        bytes = align_up(bytes);
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].size >= bytes) {
                const size_t off = free_[i].off;
                free_[i].off += bytes; free_[i].size -= bytes;
                if (!free_[i].size) free_.erase(free_.begin() + i);
                return off;
            }
        // extend the last free block if it touches the top
        if (!free_.empty() && free_.back().off + free_.back().size == top) {
            const size_t off = free_.back().off;
            top = off + bytes; free_.pop_back();
            peak = std::max(peak, top);
            return off;
        }
        const size_t off = top;
        top += bytes; peak = std::max(peak, top);
        return off;
    }
    void release(size_t off, size_t bytes) {
        bytes = align_up(bytes);
        free_.push_back({off, bytes});
        std::sort(free_.begin(), free_.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        std::vector<Blk> m;
        for (auto& b : free_) {
            if (!m.empty() && m.back().off + m.back().size == b.off) m.back().size += b.size;
            else m.push_back(b);
        }
        free_.swap(m);
    }
};

}  // namespace

struct gdt_net {
    std::vector<Op> ops;
    std::vector<Tensor> tensors;
    std::vector<int> out_ops;               // op index per external output slot
    std::vector<unsigned char> host_blob;   // packed weights / biases staged on the host until finalize
    char* dev_blob = nullptr;
    size_t zeros_off = 0;
    bool finalized = false;
    bool kcat_built = false;                // build_kcat_weights has run (at finalize, or earlier for a plan query on a graph that is not finalized yet)
    int input_op = -1;
    int precision = 0;                      // 0: fp16 activations, single MFMA pass; 1: "f16x3" (fp32 activations, split operands);
    bool head_comp = false;                 // precision mode 3: f16c with the generator head compensated too (conv_head7.hip MX pass)
                                            // 2: "f16c" (fp32 activations, fp16 product + block-scaled fp4 x fp6 correction product where a
                                            //    compensated kernel exists, f16x3 kernels elsewhere)
    size_t esize() const { return precision ? sizeof(float) : sizeof(f16); }
    // optional per-op timing (bench.py roofline): HIP events recorded on the caller's stream around every op
    bool profiling = false;
    std::vector<hipEvent_t> events;
    std::vector<double> last_flops;
    std::vector<double> last_bytes;         // algorithmic HBM bytes per op (op_bytes), merged like last_flops when ops are fused
    std::vector<int> last_variant;          // kernel variant per conv op (see gdt_launch_conv)
    float group_factor = 1.f;               // planner hint (gdt_net_set_group_factor): the geometry planned next runs concurrently with others; (their pixels + its own) / its own
    int last_joined = 0, last_level_launches = 0;   // gdt_net_forward_levels: ops whose levels shared ONE launch / launches handed back by the levels in total

    size_t blob_append(const void* data, size_t bytes) {
        const size_t off = align_up(host_blob.size());
        host_blob.resize(off + bytes);
        if (data) memcpy(host_blob.data() + off, data, bytes);
        else memset(host_blob.data() + off, 0, bytes);
        return off;
    }
    int new_tensor(int C, int Creal) { tensors.push_back(Tensor{C, Creal}); return (int)tensors.size() - 1; }
};

namespace {

struct Step {
    int op; size_t aux_off[8]; bool fused_stats; int tiles_per_image;
    int norm_into;   // INORM: index of the conv op that applies this normalisation while staging its input (-1: own apply pass)
    int norm_from;   // CONV: index of the INORM op folded into the input staging (-1: none)
    bool wb;         // INORM folded into a conv that also writes the normalised tensor out (residual / further consumers)
    bool ctf;        // CONV (transposed): runs as the single fused-phase launch
    bool aug;        // INPUT / CONV (f16c): the image is packed as augmented fp16 pixel words for the stem kernel's f16c form
    bool s2;         // CONV (stride 2, f16c): runs as the shift form over the virtual space-to-depth input
    bool ctp;        // CONV (transposed, f16x3, 64 output channels): two paired-phase launches (Op::pairs) instead of four phase launches
    int pool_into;   // CONV: index of the MAXPOOL(2,2) op whose output this conv writes directly (-1: none)
    bool skip;       // MAXPOOL fused into its producer; CONV: second / third conv of a fused Bottleneck (done by the first one's launch)
    bool bneck;      // CONV: first conv of a Bottleneck that runs as ONE launch (conv_bneck.hip): ops i, i + 1, i + 2 (identity shortcut) ...
    int bneck_ds;    // ... or {reduce, 1x1 projection shortcut} in either order at i, i + 1 (bneck_a / bneck_ds), i + 2 (3x3), i + 3 (expand + shortcut); -1: identity form
    int bneck_a;     // index of the block's reduce conv (identity form: the step itself)
    bool xexp;       // CONV (3x3): the block's expand conv (op i + 1: 1x1 + residual + ReLU) runs in the same launch on the LDS-resident tile (conv3x3_expand_rb.hip)
    int xchain;      // ... and the NEXT block's reduce conv (op index; -1: none -- 1x1, C -> 256, ReLU, reading the expand's output) as a third phase of that launch
    bool kcat;       // CONV: expand conv that also computes its projection shortcut (Op::kcat_ds, whose own step is skipped)
    bool direct;     // INPUT + its only consumer, the ResNet stem conv: the conv reads the caller's fp32 NCHW image itself when no resize is asked (conv_stem_pair_kernel)
    int stats_sets;  // CONV with fused statistics: record sets the INORM finalize sums (phase launches, or N tiles of the fused form)
};
struct Plan { std::vector<Step> steps; size_t peak = 0; };

int conv_out_dim(const gdt_conv_desc& c, int in, int k) {
    if (c.transposed) return in * 2;
    const int span = in + 2 * c.pad - k;
    return span < 0 ? 0 : span / c.stride + 1;          // floor semantics; 0 = empty (rejected by the planner)
}

// geometry part of a conv launch (everything but the pointers) for one phase of op `o` reading a tensor of size ti
void conv_geometry(const gdt_net* net, const Op& o, const PackedPhase& ph, int n, const Tensor& ti, ConvLaunch& d) {
    d.N = n; d.H = ti.H; d.W = ti.W; d.Cin = o.cin_pad; d.lc8 = ilog2(o.cin_pad / 8);
    d.Cout = o.cd.cout; d.CoutPad = o.cout_pad;
    d.OH = conv_out_dim(o.cd, ti.H, o.cd.kh); d.OW = conv_out_dim(o.cd, ti.W, o.cd.kw);
    d.pad_reflect = o.cd.pad_reflect; d.relu = o.cd.relu; d.act = o.cd.act;
    d.Kpad = ph.Kpad; d.nk = ph.Kpad / (net->precision ? 32 : 64);
    d.ntaps = ph.ntaps; d.TW = ph.TW; d.invTW = (65536 + ph.TW - 1) / ph.TW;
    d.dy0 = ph.dy0; d.dys = ph.dys; d.dx0 = ph.dx0; d.dxs = ph.dxs;
    if (o.cd.transposed) {
        d.OHg = ti.H; d.OWg = ti.W; d.sy = d.sx = 1; d.osy = d.osx = 2; d.ooy = ph.ooy; d.oox = ph.oox;
    } else {
        d.OHg = d.OH; d.OWg = d.OW; d.sy = d.sx = o.cd.stride; d.osy = d.osx = 1; d.ooy = d.oox = 0;
    }
    d.M = n * d.OHg * d.OWg;
}

bool conv_fuses_stats(const Op& o, const Tensor& ti) {      // InstanceNorm partial statistics from the conv epilogue
    if (o.stats_for < 0 || o.cd.relu || o.res >= 0) return false;
    const int hwg = o.cd.transposed ? ti.H * ti.W : conv_out_dim(o.cd, ti.H, o.cd.kh) * conv_out_dim(o.cd, ti.W, o.cd.kw);
    return hwg % 128 == 0;
}

// the fused-phase form of a transposed conv (see Op::ctf)
void ctf_geometry(const gdt_net* net, const Op& o, int n, const Tensor& ti, ConvLaunch& d) {
    conv_geometry(net, o, o.ctf, n, ti, d);
    d.Cout = d.CoutPad = 4 * o.cd.cout; d.phase_cout = o.cd.cout;
    d.OHg = ti.H; d.OWg = ti.W; d.sy = d.sx = 1; d.osy = d.osx = 2; d.ooy = d.oox = 0; d.pad_reflect = 0;
    d.M = n * d.OHg * d.OWg;
}

// the stride-2 shift form (see Op::s2): virtual channel count, 4 shift "taps", real H / W in, output grid as the patch grid
void s2_geometry(const gdt_net* net, const Op& o, int n, const Tensor& ti, ConvLaunch& d) {
    conv_geometry(net, o, o.s2, n, ti, d);
    d.Cin = 4 * o.cin_pad; d.lc8 = ilog2(o.cin_pad / 8) + 2;
    d.CoutPad = o.s2_cout_pad; d.pad_reflect = 0;
}

// a paired-phase launch of a transposed conv (see Op::pairs): 128 GEMM columns = 2 phases x 64 channels
void pair_geometry(const gdt_net* net, const Op& o, const PackedPhase& pp, int n, const Tensor& ti, ConvLaunch& d) {
    conv_geometry(net, o, pp, n, ti, d);
    d.CoutPad = 128; d.pair_cout = 64; d.ooy2 = pp.ooy2; d.oox2 = pp.oox2;
}

// shape inference, fusion decisions and workspace layout for one geometry; fills tensors[*].{H,W,off,bytes}
// direct_ok: the call does not resize its input (forward knows; the size queries plan the general case, whose footprint is the larger one)
int make_plan(gdt_net* net, int N, int RH, int RW, Plan& plan, bool direct_ok = false) {
    auto& T = net->tensors;
    const auto& ops = net->ops;
    const int nops = (int)ops.size();
    for (auto& t : T) { t.H = t.W = 0; t.last_use = -1; t.off = 0; t.bytes = 0; }
    plan.steps.assign(nops, Step{});
    for (int i = 0; i < nops; ++i) { plan.steps[i].op = i; plan.steps[i].norm_into = plan.steps[i].norm_from = -1; plan.steps[i].ctf = false; plan.steps[i].s2 = false; plan.steps[i].ctp = false; plan.steps[i].aug = false; plan.steps[i].stats_sets = 1; plan.steps[i].pool_into = -1; plan.steps[i].skip = false; plan.steps[i].bneck = false; plan.steps[i].bneck_ds = -1; plan.steps[i].bneck_a = i; plan.steps[i].kcat = false; plan.steps[i].direct = false; plan.steps[i].xexp = false; plan.steps[i].xchain = -1; }

    // ---- pass 1: shapes
    for (int i = 0; i < nops; ++i) {
        const Op& o = ops[i];
        int h = 0, w = 0;
        switch (o.kind) {
            case OP_INPUT: h = RH; w = RW; break;
            case OP_CONV: {
                const Tensor& ti = T[o.in];
                h = conv_out_dim(o.cd, ti.H, o.cd.kh); w = conv_out_dim(o.cd, ti.W, o.cd.kw);
                if (o.cd.pad_reflect) GDT_REQUIRE(o.cd.pad < ti.H && o.cd.pad < ti.W, "reflection padding needs pad < input size");
                GDT_REQUIRE(h > 0 && w > 0, "layer output would be empty for this input size");
                if (o.res >= 0) GDT_REQUIRE(T[o.res].H == h && T[o.res].W == w, "residual shape mismatch");
                break;
            }
            case OP_INORM: h = T[o.in].H; w = T[o.in].W; break;
            case OP_MAXPOOL:
                h = T[o.in].H + 2 * o.p - o.k < 0 ? 0 : (T[o.in].H + 2 * o.p - o.k) / o.s + 1;
                w = T[o.in].W + 2 * o.p - o.k < 0 ? 0 : (T[o.in].W + 2 * o.p - o.k) / o.s + 1;
                break;
            default: break;
        }
        if (o.out >= 0) {
            GDT_REQUIRE(h > 0 && w > 0, "layer output would be empty for this input size");
            T[o.out].H = h; T[o.out].W = w;
        }
    }

    // transposed convs: the single fused-phase launch when conv_igemm_rb.hip takes it
    for (int i = 0; i < nops; ++i) {
        const Op& o = ops[i];
        if (o.kind != OP_CONV) continue;
        plan.steps[i].stats_sets = (int)o.phases.size();
        if (!o.cd.transposed || !o.has_ctf || o.cd.out_f32_nchw) continue;
        ConvLaunch d{};
        ctf_geometry(net, o, N, T[o.in], d);
        d.w_frag = (const f16*)net; d.out = (f16*)net;                                 // non-null markers only
        d.stats = conv_fuses_stats(o, T[o.in]) ? (float*)net : nullptr;
        if (net->precision == 2) {                 // f16c: the compensated LDS-resident form whenever eligible
            d.w_frag = nullptr; d.w_cfrag = d.wmx_a = d.wmx_b = d.wmx_s = net;
            if (o.ctf.has_mx && gdt_conv_halo_c_ct_eligible(d)) { plan.steps[i].ctf = true; plan.steps[i].stats_sets = 1; }
            continue;
        }
        // GDT_CONV_CTF: 0 never, 1 (default) the LDS-resident kernel whenever eligible and the generic persistent GEMM only where
        // it lets the producer's InstanceNorm be folded in, 2 whenever eligible
        static const int ctf_mode = [] { const char* e = getenv("GDT_CONV_CTF"); return e ? atoi(e) : 1; }();
        bool want = ctf_mode == 2 || (ctf_mode == 1 && gdt_conv_halo_ct_eligible(d));       // the LDS-resident form always pays
        if (ctf_mode == 1 && !want) {                      // is the input an InstanceNorm (without residual) consumed only here?
            for (int j = 0; j < i; ++j)
                if (ops[j].kind == OP_INORM && ops[j].out == o.in) {
                    int uses = 0;
                    for (int k = 0; k < nops; ++k) uses += (ops[k].in == o.in) + (ops[k].res == o.in);
                    ConvLaunch dn = d; dn.in_norm = (const float*)net;
                    want = uses == 1 && (gdt_conv_igemm_rb_eligible(dn) || gdt_conv_halo_ct_eligible(dn));
                }
        }
        // record sets the finalize kernel sums: one per phase pair
        if (want && (gdt_conv_igemm_rb_eligible(d) || gdt_conv_halo_ct_eligible(d))) { plan.steps[i].ctf = true; plan.steps[i].stats_sets = 2; }
    }
    // f16c stem: the image as augmented fp16 pixel words.  Round 5: the exact split mode (f16x3) takes the same kernel -- its result is fp32-class (both rounding residuals of the
    // activation and 18-19 bits of every weight ride in the padding of the same MFMAs: 1e-6 of fp64, tests/test_hip_f16c.py::test_stem_c), it reads and writes the tensors of that mode
    // (fp32 NHWC) and replaces conv_igemm_x3<64> at 80 TFLOP/s
    for (int i = 0; i < nops && net->precision != 0; ++i) {
        const Op& o = ops[i];
        if (o.kind != OP_CONV || o.cd.transposed || o.rowsplit || o.phases.empty() || !o.phases[0].has_aug) continue;
        if (o.in < 0 || ops[net->input_op].out != o.in) continue;
        int uses = 0;
        for (int k = 0; k < nops; ++k) {
            uses += (ops[k].in == o.in) + (ops[k].res == o.in);
            if (ops[k].kind == OP_HED) for (int f = 0; f < 5; ++f) uses += ops[k].feats[f] == o.in;
        }
        if (uses != 1) continue;
        ConvLaunch d{};
        conv_geometry(net, o, o.phases[0], N, T[o.in], d);
        d.w_frag = d.w_frag2 = (const f16*)net; d.out = (f16*)net;                           // non-null markers only
        d.stats = conv_fuses_stats(o, T[o.in]) ? (float*)net : nullptr;
        if (gdt_conv_stem_c_eligible(d)) { plan.steps[i].aug = true; plan.steps[net->input_op].aug = true; }
    }
    for (int i = 0; i < nops && net->precision != 0; ++i) {
        const Op& o = ops[i];
        if (o.kind != OP_CONV || !o.has_s2) continue;
        ConvLaunch d{};
        s2_geometry(net, o, N, T[o.in], d);
        d.w_cfrag = d.wmx_a = d.wmx_b = d.wmx_s = net; d.out = (f16*)net;                  // non-null markers only
        d.stats = conv_fuses_stats(o, T[o.in]) ? (float*)net : nullptr;
        if (net->precision == 1) { d.w = d.w_lo = (const f16*)net; d.x3_form = 2; }
        if (net->precision == 1 ? gdt_conv_halo_x3_taps_eligible(d) : gdt_conv_halo_c_s2_eligible(d)) plan.steps[i].s2 = true;
    }
    for (int i = 0; i < nops && net->precision == 1; ++i) {
        const Op& o = ops[i];
        if (o.kind != OP_CONV || !o.has_pairs) continue;
        bool all = true;
        for (const PackedPhase& pp : o.pairs) {
            ConvLaunch d{};
            pair_geometry(net, o, pp, N, T[o.in], d);
            d.w = d.w_lo = (const f16*)net; d.out = (f16*)net;                              // non-null markers only
            d.stats = conv_fuses_stats(o, T[o.in]) ? (float*)net : nullptr;
            all = all && gdt_conv_halo_x3_taps_eligible(d);
        }
        plan.steps[i].ctp = all;
    }
    auto irb_norm_ok = [&](ConvLaunch d) { d.in_norm = (const float*)net; return gdt_conv_igemm_rb_eligible(d); };     // marker only
    // ---- pass 2: fold InstanceNorm(+ReLU) into the input staging of its only consumer when that is a halo-kernel conv
    static const bool allow_norm_fusion = [] { const char* e = getenv("GDT_NORM_FUSION"); return !e || atoi(e) != 0; }();
    std::vector<int> consumers(T.size(), 0), consumer_op(T.size(), -1);
    for (int i = 0; i < nops; ++i) {
        const Op& o = ops[i];
        auto use = [&](int t) { if (t >= 0) { ++consumers[t]; consumer_op[t] = i; } };
        use(o.in); use(o.res);
        if (o.kind == OP_HED) for (int k = 0; k < 5; ++k) use(o.feats[k]);
    }
    for (int j = 0; j < nops && allow_norm_fusion; ++j) {
        const Op& oj = ops[j];
        // f16x3: the patch kernel folds InstanceNorm (+ReLU, + residual, + write-back) while it stages; GDT_X3_NORM_FOLD=0 switches that off, 1 keeps it to the plain
        // norm (+ReLU) without residual / write-back (the round-5 first form)
        const char* x3_fold_env = getenv("GDT_X3_NORM_FOLD");      // (read per plan)
        const int x3_fold = x3_fold_env ? atoi(x3_fold_env) : 2;
        if (oj.kind != OP_INORM || (net->precision == 1 && !x3_fold)) continue;
        // plain norm(+ReLU): exactly one consumer.  norm + residual (ResnetBlock output): the tensor itself is still needed
        // later (as the next block's residual), so the consuming conv also writes it out -- every other consumer must come
        // after that conv in program order.
        const bool wb = oj.res >= 0 || consumers[oj.out] != 1;       // the normalised tensor itself must exist afterwards
        int k = consumer_op[oj.out];
        if (wb) {
            k = -1;
            for (int i = j + 1; i < nops && k < 0; ++i) {
                const Op& oi = ops[i];
                bool uses = oi.in == oj.out || oi.res == oj.out;
                if (oi.kind == OP_HED) for (int f = 0; f < 5; ++f) uses = uses || oi.feats[f] == oj.out;
                if (uses) k = i;
            }
            if (k < 0) continue;
        }
        const Op& ok = ops[k];
        if (ok.kind != OP_CONV || ok.in != oj.out || ok.res == oj.out) continue;
        if (ok.cd.transposed && !plan.steps[k].ctf) {
            // f16x3: the four sub-pixel phase launches of a transposed conv read the same input; each applies the norm while it stages (conv_igemm_x3.hip)
            bool all = net->precision == 1 && x3_fold >= 2 && !wb && oj.res < 0 && !ok.phases.empty();
            for (size_t p = 0; p < ok.phases.size() && all; ++p) {
                ConvLaunch d{};
                conv_geometry(net, ok, ok.phases[p], N, T[ok.in], d);
                d.w_lo = (const f16*)net;
                all = gdt_conv_x3_norm_eligible(d);
            }
            if (all) { plan.steps[j].norm_into = k; plan.steps[k].norm_from = j; plan.steps[j].wb = false; }
            continue;
        }
        if (ok.cd.out_f32_nchw && !ok.rowsplit) continue;
        if (plan.steps[k].ctf && net->precision == 2) {
            ConvLaunch d{};
            ctf_geometry(net, ok, N, T[ok.in], d);
            d.w_cfrag = d.wmx_a = d.wmx_b = d.wmx_s = net; d.out = (f16*)net;
            d.stats = conv_fuses_stats(ok, T[ok.in]) ? (float*)net : nullptr;
            d.in_norm = (const float*)net; d.in_res = oj.res >= 0 ? (const f16*)net : nullptr;
            if (consumers[oj.out] == 1 && gdt_conv_halo_c_ct_eligible(d)) { plan.steps[j].norm_into = k; plan.steps[k].norm_from = j; plan.steps[j].wb = false; }
            continue;
        }
        if (plan.steps[k].s2) {
            if (oj.res >= 0) continue;
            ConvLaunch d{};
            s2_geometry(net, ok, N, T[ok.in], d);
            d.w_cfrag = d.wmx_a = d.wmx_b = d.wmx_s = net; d.out = (f16*)net;
            d.stats = conv_fuses_stats(ok, T[ok.in]) ? (float*)net : nullptr;
            d.in_norm = (const float*)net; d.in_out = wb ? (f16*)net : nullptr;
            if (net->precision == 1) {         // conv3x3_halo_x3.hip FORM 2: plain norm (+ReLU)
                d.w = d.w_lo = (const f16*)net; d.x3_form = 2;
                if (x3_fold >= 2 && !wb && gdt_conv_halo_x3_taps_eligible(d)) { plan.steps[j].norm_into = k; plan.steps[k].norm_from = j; plan.steps[j].wb = false; }
                continue;
            }
            if (gdt_conv_halo_c_s2_eligible(d)) { plan.steps[j].norm_into = k; plan.steps[k].norm_from = j; plan.steps[j].wb = wb; }
            continue;
        }
        if (net->precision != 0 && ok.rowsplit) {  // f16c / f16x3 head: the fused 7x7 kernel normalises while it stages its fp32 input
            ConvLaunch h{};
            conv_geometry(net, ok, ok.phases[0], N, T[ok.in], h);
            h.w_frag = ok.phases[0].has_frag ? (const f16*)net : nullptr; h.out_f32 = (float*)net; h.Cout = ok.cd.cout;
            if (!wb && oj.res < 0 && gdt_conv_head7_eligible(h)) { plan.steps[j].norm_into = k; plan.steps[k].norm_from = j; plan.steps[j].wb = false; }
            continue;
        }
        if (net->precision == 2) {                 // f16c: the compensated halo kernel folds norm (+ReLU, +residual, +write-back)
            if (!ok.phases[0].has_mx) continue;
            ConvLaunch d{};
            conv_geometry(net, ok, ok.phases[0], N, T[ok.in], d);
            d.w = (const f16*)net; d.w_cfrag = d.wmx_a = d.wmx_b = d.wmx_s = net; d.out = (f16*)net;         // non-null markers only
            d.stats = conv_fuses_stats(ok, T[ok.in]) ? (float*)net : nullptr;
            d.in_norm = (const float*)net; d.in_res = oj.res >= 0 ? (const f16*)net : nullptr; d.in_out = wb ? (f16*)net : nullptr;
            if (gdt_conv_halo_c_eligible(d)) { plan.steps[j].norm_into = k; plan.steps[k].norm_from = j; plan.steps[j].wb = wb; }
            continue;
        }
        ConvLaunch d{};
        if (plan.steps[k].ctf) {
            ctf_geometry(net, ok, N, T[ok.in], d);
            d.w_frag = (const f16*)net; d.out = (f16*)net;
            d.stats = conv_fuses_stats(ok, T[ok.in]) ? (float*)net : nullptr;
            // (a residual without further consumers needs no write-back: the LDS-resident form adds it while staging)
            ConvLaunch dn = d; dn.in_norm = (const float*)net; dn.in_res = oj.res >= 0 ? (const f16*)net : nullptr;
            const bool ok_fold = consumers[oj.out] == 1 && (gdt_conv_halo_ct_eligible(dn) || (oj.res < 0 && gdt_conv_igemm_rb_eligible(dn)));
            if (ok_fold) { plan.steps[j].norm_into = k; plan.steps[k].norm_from = j; plan.steps[j].wb = false; }
            continue;
        }
        conv_geometry(net, ok, ok.phases[0], N, T[ok.in], d);
        if (ok.rowsplit) { d.Cout = ok.rs_cout8; d.out_f32 = nullptr; }
        d.w_lo = net->precision ? (const f16*)net : nullptr;                       // non-null marker only
        d.w_frag = ok.phases[0].has_frag ? (const f16*)net : nullptr;              // non-null marker only
        d.out = (ok.cd.out_f32_nchw || ok.rowsplit) ? nullptr : (f16*)net;         // non-null marker only
        d.stats = conv_fuses_stats(ok, T[ok.in]) ? (float*)net : nullptr;          // non-null marker only
        d.out_f32 = (ok.cd.out_f32_nchw && !ok.rowsplit) ? (float*)net : nullptr;
        const bool fold = net->precision ? (((x3_fold >= 2 || (!wb && oj.res < 0)) && gdt_conv_halo_x3_eligible(d)) ||
                                            (x3_fold >= 2 && !wb && oj.res < 0 && !ok.rowsplit && !ok.cd.out_f32_nchw && gdt_conv_x3_norm_eligible(d)))
                                         : (gdt_conv_halo_eligible(d) || (!wb && (gdt_conv_igemm_norm_eligible(d) || irb_norm_ok(d))));
        if (fold) { plan.steps[j].norm_into = k; plan.steps[k].norm_from = j; plan.steps[j].wb = wb; }
        static const bool plan_dbg = getenv("GDT_PLAN_DEBUG") != nullptr;
        if (plan_dbg) fprintf(stderr, "[plan] inorm %d -> conv %d: Cin %d s%d k%d rowsplit %d fold %d\n", j, k, d.Cin, ok.cd.stride, ok.cd.kh, (int)ok.rowsplit, (int)fold);
    }

    // ---- statistics record sets of the convs that run on conv_igemm_rb.hip (same order of choice as gdt_launch_conv)
    for (int i = 0; i < nops; ++i) {
        const Op& o = ops[i];
        if (o.kind != OP_CONV || plan.steps[i].ctf || o.cd.transposed || o.rowsplit || o.cd.out_f32_nchw || net->precision) continue;
        if (!conv_fuses_stats(o, T[o.in])) continue;
        ConvLaunch d{};
        conv_geometry(net, o, o.phases[0], N, T[o.in], d);
        d.w_frag = o.phases[0].has_frag ? (const f16*)net : nullptr; d.out = (f16*)net; d.stats = (float*)net;       // markers only
        d.in_norm = plan.steps[i].norm_from >= 0 ? (const float*)net : nullptr;
        if (plan.steps[i].norm_from >= 0 && plan.steps[plan.steps[i].norm_from].wb) continue;                      // (patch kernels)
        if (!gdt_conv_stem_eligible(d) && !gdt_conv_halo_rb_eligible(d) && !gdt_conv_halo_eligible(d) && gdt_conv_igemm_rb_eligible(d))
            plan.steps[i].stats_sets = gdt_conv_igemm_rb_stats_sets(d);
    }

    // ---- pass 2b: MaxPool2d(2, 2) fused into the epilogue of its producer (VGG16 stages): conv -> pool with no other consumer
    for (int j = 0; j < nops; ++j) {
        const Op& oj = ops[j];
        if (oj.kind != OP_MAXPOOL || oj.k != 2 || oj.s != 2 || oj.p != 0 || net->precision || consumers[oj.in] != 1) continue;
        int i = -1;
        for (int k = 0; k < j; ++k) if (ops[k].kind == OP_CONV && ops[k].out == oj.in) i = k;
        if (i < 0 || ops[i].cd.transposed || ops[i].cd.out_f32_nchw || ops[i].res >= 0) continue;
        ConvLaunch d{};
        conv_geometry(net, ops[i], ops[i].phases[0], N, T[ops[i].in], d);
        d.w_frag = ops[i].phases[0].has_frag ? (const f16*)net : nullptr; d.out = (f16*)net;            // non-null markers only
        d.in_norm = plan.steps[i].norm_from >= 0 ? (const float*)net : nullptr;
        d.stats = conv_fuses_stats(ops[i], T[ops[i].in]) ? (float*)net : nullptr;
        if (plan.steps[i].norm_from >= 0 && plan.steps[plan.steps[i].norm_from].wb) continue;
        if (gdt_conv_pool2_eligible(d)) { plan.steps[i].pool_into = j; plan.steps[j].skip = true; }
    }

    // ---- pass 2c (fp16 mode): identity Bottlenecks as one launch -- conv 1x1 (C -> MID, ReLU) -> conv 3x3 s1 p1 (MID -> MID, ReLU) -> conv 1x1
    // (MID -> C) + residual = the first conv's input, ReLU; the two intermediate tensors have no other consumer and are never allocated
    for (int i = 0; i + 2 < nops && !net->precision; ++i) {
        const Op &a = ops[i], &b = ops[i + 1], &c = ops[i + 2];
        if (a.kind != OP_CONV || b.kind != OP_CONV || c.kind != OP_CONV) continue;
        auto plain = [&](const Op& o) { return !o.cd.transposed && !o.cd.out_f32_nchw && !o.rowsplit && o.stats_for < 0 && o.cd.stride == 1 && o.phases.size() == 1 && o.phases[0].has_frag && o.has_bias; };
        if (!plain(a) || !plain(b) || !plain(c)) continue;
        if (a.cd.kh != 1 || a.cd.kw != 1 || a.cd.pad != 0 || !a.cd.relu || a.res >= 0) continue;
        if (b.cd.kh != 3 || b.cd.kw != 3 || b.cd.pad != 1 || b.cd.pad_reflect || !b.cd.relu || b.res >= 0 || b.in != a.out) continue;
        if (c.cd.kh != 1 || c.cd.kw != 1 || c.cd.pad != 0 || !c.cd.relu || c.in != b.out || c.res != a.in) continue;
        if (consumers[a.out] != 1 || consumers[b.out] != 1) continue;
        if (plan.steps[i].norm_from >= 0 || plan.steps[i + 1].norm_from >= 0 || plan.steps[i + 2].norm_from >= 0) continue;
        if (plan.steps[i].pool_into >= 0 || plan.steps[i + 1].pool_into >= 0 || plan.steps[i + 2].pool_into >= 0) continue;
        const int C = a.cd.cin, mid = a.cd.cout;
        if (a.cin_pad != C || a.cout_pad != mid || b.cd.cin != mid || b.cd.cout != mid || b.cout_pad != mid || c.cd.cin != mid || c.cd.cout != C || c.cout_pad != C) continue;
        if (!gdt_bneck_eligible(C, C, mid, N, T[a.in].H, T[a.in].W)) continue;
        plan.steps[i].bneck = true; plan.steps[i + 1].skip = true; plan.steps[i + 2].skip = true;
    }
    // ... and the projection-shortcut form (first block of a stage at stride 1): conv 1x1 (CIN -> MID, ReLU), conv 1x1 (CIN -> C, no ReLU: the shortcut,
    // same input), conv 3x3, conv 1x1 (MID -> C) + shortcut, ReLU
    for (int i = 0; i + 3 < nops && !net->precision; ++i) {
        if (ops[i].kind != OP_CONV || ops[i + 1].kind != OP_CONV) continue;
        const bool ds_first = !ops[i].cd.relu;                   // (the shortcut projection has no ReLU; engine.py emits it before the reduce conv)
        const int ia = ds_first ? i + 1 : i, ids = ds_first ? i : i + 1;
        const Op &a = ops[ia], &ds = ops[ids], &b = ops[i + 2], &c = ops[i + 3];
        if (a.kind != OP_CONV || ds.kind != OP_CONV || b.kind != OP_CONV || c.kind != OP_CONV) continue;
        auto plain = [&](const Op& o) { return !o.cd.transposed && !o.cd.out_f32_nchw && !o.rowsplit && o.stats_for < 0 && o.cd.stride == 1 && o.phases.size() == 1 && o.phases[0].has_frag && o.has_bias; };
        if (!plain(a) || !plain(ds) || !plain(b) || !plain(c)) continue;
        if (a.cd.kh != 1 || a.cd.kw != 1 || a.cd.pad != 0 || !a.cd.relu || a.res >= 0) continue;
        if (ds.cd.kh != 1 || ds.cd.kw != 1 || ds.cd.pad != 0 || ds.cd.relu || ds.res >= 0 || ds.in != a.in) continue;
        if (b.cd.kh != 3 || b.cd.kw != 3 || b.cd.pad != 1 || b.cd.pad_reflect || !b.cd.relu || b.res >= 0 || b.in != a.out) continue;
        if (c.cd.kh != 1 || c.cd.kw != 1 || c.cd.pad != 0 || !c.cd.relu || c.in != b.out || c.res != ds.out) continue;
        if (consumers[a.out] != 1 || consumers[b.out] != 1 || consumers[ds.out] != 1) continue;
        bool folded = false;
        for (int k = i; k < i + 4; ++k) folded = folded || plan.steps[k].norm_from >= 0 || plan.steps[k].pool_into >= 0 || plan.steps[k].skip || plan.steps[k].bneck;
        if (folded) continue;
        const int cin = a.cd.cin, mid = a.cd.cout, C = c.cd.cout;
        if (a.cin_pad != cin || a.cout_pad != mid || ds.cd.cin != cin || ds.cin_pad != cin || ds.cd.cout != C || ds.cout_pad != C || b.cd.cin != mid || b.cd.cout != mid ||
            b.cout_pad != mid || c.cd.cin != mid || c.cout_pad != C || cin == C) continue;
        if (!gdt_bneck_eligible(cin, C, mid, N, T[a.in].H, T[a.in].W)) continue;
        plan.steps[i].bneck = true; plan.steps[i].bneck_ds = ids; plan.steps[i].bneck_a = ia;
        plan.steps[i + 1].skip = true; plan.steps[i + 2].skip = true; plan.steps[i + 3].skip = true;
    }

    // ---- pass 2c'' (fp16 mode): Bottlenecks that did not fuse as a whole (MID = 256: ResNet-101 layer3): 3x3 conv + expand conv + residual as one launch; the
    // 3x3's output tensor has no other consumer and is never allocated
    const char* xexp_env = getenv("GDT_CONV_XEXP");                  // 0: off (read when a net plans a geometry: A/B inside one process)
    for (int i = 0; i + 1 < nops && !net->precision && !(xexp_env && atoi(xexp_env) == 0); ++i) {
        const Op &b = ops[i], &c = ops[i + 1];
        if (b.kind != OP_CONV || c.kind != OP_CONV) continue;
        auto plain = [&](const Op& o) { return !o.cd.transposed && !o.cd.out_f32_nchw && !o.rowsplit && o.stats_for < 0 && o.cd.stride == 1 && o.phases.size() == 1 && o.phases[0].has_frag && o.has_bias; };
        if (!plain(b) || !plain(c)) continue;
        if (b.cd.kh != 3 || b.cd.kw != 3 || b.cd.pad != 1 || b.cd.pad_reflect || !b.cd.relu || b.res >= 0) continue;
        if (c.cd.kh != 1 || c.cd.kw != 1 || c.cd.pad != 0 || !c.cd.relu || c.in != b.out || c.res < 0 || c.kcat_ds >= 0 || !c.has_bias_frag) continue;
        if (consumers[b.out] != 1) continue;
        bool folded = false;
        for (int k = i; k < i + 2; ++k) folded = folded || plan.steps[k].norm_from >= 0 || plan.steps[k].pool_into >= 0 || plan.steps[k].skip || plan.steps[k].bneck;
        if (folded) continue;
        if (b.cd.cin != 256 || b.cin_pad != 256 || b.cd.cout != 256 || b.cout_pad != 256 || c.cd.cin != 256 || c.cout_pad != c.cd.cout) continue;
        ConvLaunch d{};
        conv_geometry(net, b, b.phases[0], N, T[b.in], d);
        d.w_frag = (const f16*)net; d.x_w_frag = (const f16*)net; d.bias = (const float*)net; d.x_bias = (const f16*)net;        // non-null markers only
        d.res = (const f16*)net; d.out = (f16*)net; d.x_cout = c.cd.cout; d.relu = 1;
        d.group_factor = net->group_factor;
        if (!gdt_conv3x3_expand_eligible(d)) continue;
        plan.steps[i].xexp = true; plan.steps[i + 1].skip = true;
        // ... chained with the next block's reduce conv (torchvision Bottleneck.conv1 of the following block): 1x1, stride 1, C -> 256, bias, ReLU, no residual,
        // reading the tensor this launch writes; its own launch (and its read of that tensor from HBM) goes away
        int j2 = i + 2;
        while (j2 < nops && ops[j2].kind == OP_OUT_NCHW) ++j2;          // (feature taps between the blocks read tensors this launch has written: they stay in place)
        if (j2 < nops) {
            const Op& a2 = ops[j2];
            const bool ok = a2.kind == OP_CONV && plain(a2) && a2.cd.kh == 1 && a2.cd.kw == 1 && a2.cd.pad == 0 && a2.cd.relu && a2.res < 0 && a2.in == c.out &&
                            a2.cd.cin == c.cd.cout && a2.cin_pad == a2.cd.cin && a2.cd.cout == 256 && a2.cout_pad == 256 && a2.kcat_ds < 0 && a2.out >= 0 &&
                            plan.steps[j2].norm_from < 0 && plan.steps[j2].pool_into < 0 && !plan.steps[j2].skip && !plan.steps[j2].bneck && !plan.steps[j2].kcat;
            if (ok && gdt_conv3x3_expand_chain_eligible(d)) { plan.steps[i].xchain = j2; plan.steps[j2].skip = true; }
        }
    }

    // ---- pass 2d (fp16 mode): projection shortcut folded into the expand conv (K-concatenated 1x1, conv1x1_rb.hip) where the block did not fuse as a whole
    for (int i = 0; i < nops && !net->precision; ++i) {
        const Op& c = ops[i];
        if (c.kind != OP_CONV || c.kcat_ds < 0) continue;
        const int ids = c.kcat_ds;
        const Op& ds = ops[ids];
        if (plan.steps[i].skip || plan.steps[i].bneck || plan.steps[ids].skip || plan.steps[ids].bneck) continue;
        if (plan.steps[i].norm_from >= 0 || plan.steps[ids].norm_from >= 0 || plan.steps[i].pool_into >= 0 || consumers[ds.out] != 1) continue;
        ConvLaunch d{};
        conv_geometry(net, c, c.phases[0], N, T[c.in], d);
        d.w_frag = (const f16*)net; d.out = (f16*)net; d.in2 = (const f16*)net;            // non-null markers only
        d.Kpad = c.cin_pad + ds.cin_pad; d.in2_cin = ds.cin_pad; d.in2_h = T[ds.in].H; d.in2_w = T[ds.in].W; d.in2_stride = ds.cd.stride;
        if (!gdt_conv_1x1_cat_eligible(d)) continue;
        plan.steps[i].kcat = true; plan.steps[ids].skip = true;
    }

    // ---- pass 2e (fp16 mode): the ResNet stem straight from the caller's image (no input pack) -- decided here, taken by the executor when the call does not resize
    if (direct_ok && !net->precision && nops >= 2 && ops[0].kind == OP_INPUT && ops[0].in_c <= 3 && consumers[ops[0].out] == 1) {
        const int j = consumer_op[ops[0].out];
        const Op& o = ops[j];
        if (o.kind == OP_CONV && o.in == ops[0].out && o.res < 0 && o.phases.size() == 1 && o.phases[0].has_pair && !plan.steps[j].aug && plan.steps[j].norm_from < 0 &&
            plan.steps[j].pool_into < 0 && o.stats_for < 0 && !plan.steps[j].skip && !plan.steps[j].bneck) {
            ConvLaunch d{};
            conv_geometry(net, o, o.phases[0], N, T[o.in], d);
            d.w_frag = (const f16*)net; d.out = (f16*)net;                                 // non-null markers only
            if (gdt_conv_stem_pair_eligible(d)) {
                plan.steps[0].direct = true; plan.steps[j].direct = true;
                // ... and the MaxPool2d(3, 2, 1) behind it, when it is the stem's only consumer: the stem launch writes the pooled tensor
                static const bool pool_ok = [] { const char* e = getenv("GDT_CONV_STEM_POOL"); return !e || atoi(e) != 0; }();
                const int jp = consumers[o.out] == 1 ? consumer_op[o.out] : -1;
                if (pool_ok && jp >= 0 && ops[jp].kind == OP_MAXPOOL && ops[jp].k == 3 && ops[jp].s == 2 && ops[jp].p == 1 && o.cd.relu && o.slot < 0) {
                    plan.steps[j].pool_into = jp; plan.steps[jp].skip = true;
                }
            }
        }
    }

    // ---- pass 3: liveness + first-fit layout
    auto conv_input = [&](int i) { return plan.steps[i].norm_from >= 0 ? ops[plan.steps[i].norm_from].in : ops[i].in; };
    for (int i = 0; i < nops; ++i) {
        const Op& o = ops[i];
        const int in = o.kind == OP_CONV ? conv_input(i) : o.in;
        if (in >= 0) T[in].last_use = i;
        if (o.res >= 0) T[o.res].last_use = i;
        if (o.kind == OP_CONV && plan.steps[i].kcat) T[ops[o.kcat_ds].in].last_use = i;      // the expand conv reads the projection's input itself
        if (o.kind == OP_CONV && plan.steps[i].norm_from >= 0) {
            const Op& nj = ops[plan.steps[i].norm_from];          // the conv reads the residual and (wb) writes the norm's output tensor
            if (nj.res >= 0) T[nj.res].last_use = std::max(T[nj.res].last_use, i);
            if (plan.steps[plan.steps[i].norm_from].wb) T[nj.out].last_use = std::max(T[nj.out].last_use, i);
        }
        if (o.kind == OP_HED) for (int k = 0; k < 5; ++k) T[o.feats[k]].last_use = i;
    }
    Arena arena;
    std::vector<size_t> slab_off(nops, 0), slab_bytes(nops, 0);
    std::vector<std::vector<std::pair<size_t, size_t>>> deferred(nops);      // releases to perform after op i
    for (int i = 0; i < nops; ++i) {
        const Op& o = ops[i];
        Step& st = plan.steps[i];
        auto alloc_out = [&]() {
            Tensor& t = T[o.out];
            t.bytes = (size_t)N * t.H * t.W * t.C * net->esize();
            t.off = arena.alloc(t.bytes);
        };
        switch (o.kind) {
            case OP_INPUT: if (!st.direct) alloc_out(); break;      // (direct: the stem conv reads the caller's image, the packed tensor never exists)
            case OP_CONV: {
                const Tensor& ti = T[o.in];       // same size as the raw tensor when the norm is folded
                const int oh = conv_out_dim(o.cd, ti.H, o.cd.kh);
                if (st.skip) break;                           // (fused Bottleneck: done by the block's first conv)
                if (st.bneck) {                               // the launch writes the block output; r and t (and the projected shortcut) never exist
                    Tensor& t = T[ops[i + (st.bneck_ds >= 0 ? 3 : 2)].out];
                    t.bytes = (size_t)N * t.H * t.W * t.C * net->esize();
                    t.off = arena.alloc(t.bytes);
                    break;
                }
                if (st.xexp) {                                // the launch writes the expand conv's output; the 3x3's own output never exists
                    Tensor& t = T[ops[i + 1].out];
                    t.bytes = (size_t)N * t.H * t.W * t.C * net->esize();
                    t.off = arena.alloc(t.bytes);
                    if (st.xchain >= 0) {                     // ... and the next block's reduce output
                        Tensor& t2 = T[ops[st.xchain].out];
                        t2.bytes = (size_t)N * t2.H * t2.W * t2.C * net->esize();
                        t2.off = arena.alloc(t2.bytes);
                    }
                    break;
                }
                if (st.pool_into >= 0) {                      // the conv writes the pooled tensor; its own output never exists
                    Tensor& t = T[ops[st.pool_into].out];
                    t.bytes = (size_t)N * t.H * t.W * t.C * net->esize();
                    t.off = arena.alloc(t.bytes);
                } else if (o.out >= 0) alloc_out();
                if (o.rowsplit) {
                    const size_t b = (size_t)N * oh * ti.W * o.rs_cout8 * net->esize();
                    st.aux_off[1] = arena.alloc(b);
                    arena.release(st.aux_off[1], b);
                }
                if (conv_fuses_stats(o, ti)) {
                    const int hwg = o.cd.transposed ? ti.H * ti.W : oh * conv_out_dim(o.cd, ti.W, o.cd.kw);
                    const size_t tiles = (size_t)st.stats_sets * N * (hwg / 128);
                    st.fused_stats = true; st.tiles_per_image = hwg / 128;
                    slab_bytes[i] = tiles * 2 * o.cd.cout * sizeof(float);
                    st.aux_off[0] = arena.alloc(slab_bytes[i]);
                    slab_off[i] = st.aux_off[0];
                }
                break;
            }
            case OP_INORM: {
                const Tensor& ti = T[o.in];
                const size_t mr_bytes = (size_t)N * ti.C * 2 * sizeof(float);
                if (st.norm_into < 0 || st.wb) alloc_out();               // (folded with write-back: the consuming conv writes it)
                st.aux_off[1] = arena.alloc(mr_bytes);
                if (o.stats_from >= 0 && slab_bytes[o.stats_from]) {
                    st.fused_stats = true;
                    st.tiles_per_image = plan.steps[o.stats_from].tiles_per_image;
                    st.aux_off[0] = slab_off[o.stats_from];
                    arena.release(slab_off[o.stats_from], slab_bytes[o.stats_from]);
                } else {
                    const int chunks = gdt_in_stats_chunks(ti.H * ti.W);
                    st.aux_off[0] = arena.alloc((size_t)N * chunks * 2 * ti.C * sizeof(float));
                    arena.release(st.aux_off[0], (size_t)N * chunks * 2 * ti.C * sizeof(float));
                }
                if (st.norm_into >= 0) deferred[st.norm_into].push_back({st.aux_off[1], mr_bytes});   // the conv reads it
                else arena.release(st.aux_off[1], mr_bytes);
                break;
            }
            case OP_MAXPOOL: if (!st.skip) alloc_out(); break;
            case OP_GEM: {
                const Tensor& ti = T[o.in];
                st.aux_off[0] = arena.alloc((size_t)N * ti.C * sizeof(float));
                arena.release(st.aux_off[0], (size_t)N * ti.C * sizeof(float));
                break;
            }
            case OP_OUT_NCHW: break;
            case OP_HED: {
                size_t sz[5];
                for (int k = 0; k < 5; ++k) {
                    const Tensor& tf = T[o.feats[k]];
                    sz[k] = (size_t)N * tf.H * tf.W * sizeof(float);
                    st.aux_off[k] = arena.alloc(sz[k]);
                }
                for (int k = 0; k < 5; ++k) arena.release(st.aux_off[k], sz[k]);
                break;
            }
        }
        // free dead inputs
        auto maybe_free = [&](int t) {
            if (t >= 0 && T[t].last_use == i && T[t].bytes) { arena.release(T[t].off, T[t].bytes); T[t].last_use = -2; }
        };
        maybe_free(o.kind == OP_CONV ? conv_input(i) : o.in); maybe_free(o.res);
        if (o.kind == OP_CONV && st.kcat) maybe_free(ops[o.kcat_ds].in);
        if (o.kind == OP_HED) for (int k = 0; k < 5; ++k) maybe_free(o.feats[k]);
        if (o.kind == OP_CONV && st.norm_from >= 0) { maybe_free(ops[st.norm_from].res); maybe_free(ops[st.norm_from].out); }
        if (o.out >= 0 && T[o.out].last_use == -1 && T[o.out].bytes) arena.release(T[o.out].off, T[o.out].bytes);   // never consumed
        for (auto& r : deferred[i]) arena.release(r.first, r.second);
    }
    plan.peak = arena.peak;
    return GDT_OK;
}

// ---- host-side weight packing ----------------------------------------------------------------------------------
// e2m3 (OCP fp6: 1 sign, 2 exponent (bias 1), 3 mantissa bits; max 7.5, subnormal step 0.125), round to nearest even
int quant_e2m3(float x) {
    const int sign = std::signbit(x) ? 32 : 0;
    float ax = std::fabs(x);
    if (!(ax < 7.5f)) ax = 7.5f;
    if (ax < 1.f) return sign | (int)std::nearbyint(ax * 8.f);          // 0 .. 8 (8 = 1.0: exponent field 1, mantissa 0)
    int e = ax >= 4.f ? 2 : (ax >= 2.f ? 1 : 0);
    int m = (int)std::nearbyint((std::ldexp(ax, -e) - 1.f) * 8.f);
    if (m == 8) { m = 0; ++e; }
    if (e > 2) { e = 2; m = 7; }
    return sign | ((e + 1) << 3) | m;
}

// Block-scaled correction operands of a packed weight matrix wf [cout_pad][Kpad] (fp32, BatchNorm folded) in MFMA fragment order:
// per (32-channel block cb, 32-k block ms, lane): lanes 0-31 hold fp16(w), lanes 32-63 hold w - fp16(w) of output channel cb*32 + (lane & 31),
// each as 32 e2m3 values of  value * 2^-e  with the block's own E8M0 exponent byte 127 + e (largest magnitude of the block mapped into
// (3.75, 7.5]).  Element i sits at bit 6i of the lane's 24 bytes: the first 16 go to `a`, the last 8 to `b`, the E8M0 scale (a dword) to `sc` -- three
// arrays, each contiguous over the 64 lanes of a fragment, so that the kernel's dwordx4 / dwordx2 / dword loads touch 8 + 4 + 2 cache lines per
// fragment and land exactly in the MFMA's operand registers (conv3x3_halo_c.hip load_bq).
// Layouts are grouped per 128 output channels (see ConvLaunch::w_cfrag): index = ((group * steps + step) * 4 + block in group) * 64 + lane.
void pack_mx(const std::vector<float>& wf, int cout_pad, int Kpad, std::vector<unsigned char>& a, std::vector<unsigned char>& b,
             std::vector<unsigned>& sc, std::vector<f16>& wc) {
    const int ncb = cout_pad / 32, nms = Kpad / 32, nks = Kpad / 16;
    const size_t ncb4 = (size_t)(ncb + 3) / 4 * 4;            // whole groups of four 32-channel blocks (the head has a single block)
    a.assign(ncb4 * nms * 64 * 16, 0); b.assign(ncb4 * nms * 64 * 8, 0); sc.assign(ncb4 * nms * 64, 0);
    wc.assign(ncb4 * 32 * (size_t)Kpad, (f16)0.f);
    for (int cb = 0; cb < ncb; ++cb)
        for (int ks = 0; ks < nks; ++ks)
            for (int ln = 0; ln < 64; ++ln) {
                const float* src = wf.data() + (size_t)(cb * 32 + (ln & 31)) * Kpad + ks * 16 + (ln >> 5) * 8;
                f16* dst = wc.data() + ((((size_t)(cb >> 2) * nks + ks) * 4 + (cb & 3)) * 64 + ln) * 8;
                for (int e = 0; e < 8; ++e) dst[e] = (f16)src[e];
            }
    for (int cb = 0; cb < ncb; ++cb)
        for (int ms = 0; ms < nms; ++ms)
            for (int ln = 0; ln < 64; ++ln) {
                const float* src = wf.data() + (size_t)(cb * 32 + (ln & 31)) * Kpad + ms * 32;
                float v[32], mx = 0.f;
                for (int i = 0; i < 32; ++i) {
                    const float hi = (float)(f16)src[i];
                    v[i] = (ln >> 5) ? src[i] - hi : hi;
                    mx = std::max(mx, std::fabs(v[i]));
                }
                int e = 0;
                if (mx > 0.f) { e = (int)std::ceil(std::log2(mx / 7.5f)); if (std::ldexp(mx, -e) > 7.5f) ++e; }
                e = std::min(std::max(e, -126), 127);
                unsigned char bytes[24] = {0};
                for (int i = 0; i < 32; ++i) {
                    const unsigned code = (unsigned)quant_e2m3(std::ldexp(v[i], -e));
                    const int bit = 6 * i;
                    bytes[bit >> 3] |= (unsigned char)(code << (bit & 7));
                    if ((bit & 7) > 2) bytes[(bit >> 3) + 1] |= (unsigned char)(code >> (8 - (bit & 7)));
                }
                const size_t fi = (((size_t)(cb >> 2) * nms + ms) * 4 + (cb & 3)) * 64 + ln;
                const unsigned scale = (unsigned)(127 + e);           // E8M0 block scale
                memcpy(a.data() + fi * 16, bytes, 16); memcpy(b.data() + fi * 8, bytes + 16, 8); sc[fi] = scale;
            }
}

// The same operands in the fragment order of the 16 x 16 MFMA shapes (conv3x3_halo_c16.hip), grouped per 64 output channels (a wave's slice):
//   wc   index = ((group * K/32 + step) * 4 + block) * 64 + lane: lane (n = lane & 15, g = lane >> 4) = fp16(w[group * 64 + block * 16 + n][32 step + 8 g .. +7])
//   a/b index = ((group * K/64 + m) * 4 + block) * 64 + lane, sc index = ((group * K/64 + m) * 64 + lane) * 4 + block: lane (n, blk = lane >> 4) = 32 e2m3 values (+ E8M0 scale) of the 32 k-values
//   64 m + 32 (blk >> 1) .. +31: fp16(w) for blk 0 / 2, w - fp16(w) for blk 1 / 3 -- the K blocks of v_mfma_scale_f32_16x16x128_f8f6f4, which meet
//   the activation row [a_lo | a_hi | a_lo' | a_hi'] block by block.
void pack_mx16(const std::vector<float>& wf, int cout_pad, int Kpad, std::vector<unsigned char>& a, std::vector<unsigned char>& b,
               std::vector<unsigned>& sc, std::vector<f16>& wc) {
    const int ng = cout_pad / 64, nks = Kpad / 32, nms = Kpad / 64;
    a.assign((size_t)ng * nms * 4 * 64 * 16, 0); b.assign((size_t)ng * nms * 4 * 64 * 8, 0); sc.assign((size_t)ng * nms * 4 * 64, 0);
    wc.assign((size_t)cout_pad * Kpad, (f16)0.f);
    for (int g = 0; g < ng; ++g)
        for (int ks = 0; ks < nks; ++ks)
            for (int cb = 0; cb < 4; ++cb)
                for (int ln = 0; ln < 64; ++ln) {
                    const float* src = wf.data() + (size_t)(g * 64 + cb * 16 + (ln & 15)) * Kpad + ks * 32 + (ln >> 4) * 8;
                    f16* dst = wc.data() + ((((size_t)g * nks + ks) * 4 + cb) * 64 + ln) * 8;
                    for (int e = 0; e < 8; ++e) dst[e] = (f16)src[e];
                }
    for (int g = 0; g < ng; ++g)
        for (int ms = 0; ms < nms; ++ms)
            for (int cb = 0; cb < 4; ++cb)
                for (int ln = 0; ln < 64; ++ln) {
                    const int blk = ln >> 4;
                    const float* src = wf.data() + (size_t)(g * 64 + cb * 16 + (ln & 15)) * Kpad + ms * 64 + (blk >> 1) * 32;
                    float v[32], mx = 0.f;
                    for (int i = 0; i < 32; ++i) {
                        const float hi = (float)(f16)src[i];
                        v[i] = (blk & 1) ? src[i] - hi : hi;
                        mx = std::max(mx, std::fabs(v[i]));
                    }
                    int e = 0;
                    if (mx > 0.f) { e = (int)std::ceil(std::log2(mx / 7.5f)); if (std::ldexp(mx, -e) > 7.5f) ++e; }
                    e = std::min(std::max(e, -126), 127);
                    unsigned char bytes[24] = {0};
                    for (int i = 0; i < 32; ++i) {
                        const unsigned code = (unsigned)quant_e2m3(std::ldexp(v[i], -e));
                        const int bit = 6 * i;
                        bytes[bit >> 3] |= (unsigned char)(code << (bit & 7));
                        if ((bit & 7) > 2) bytes[(bit >> 3) + 1] |= (unsigned char)(code >> (8 - (bit & 7)));
                    }
                    const size_t fi = (((size_t)g * nms + ms) * 4 + cb) * 64 + ln;
                    memcpy(a.data() + fi * 16, bytes, 16); memcpy(b.data() + fi * 8, bytes + 16, 8);
                    sc[(((size_t)g * nms + ms) * 64 + ln) * 4 + cb] = (unsigned)(127 + e);      // (a lane's four block scales side by side: one dwordx4)
                }
}

// the 15 KB records of conv3x3_halo_c16.hip from a [cols][Kpad] fp32 matrix (cols % 64 == 0, Kpad % 64 == 0)
std::vector<unsigned char> pack_records16(const std::vector<float>& wf, int cols, int Kpad) {
    std::vector<unsigned char> ma, mb; std::vector<unsigned> msc; std::vector<f16> wc;
    pack_mx16(wf, cols, Kpad, ma, mb, msc, wc);
    const size_t nrec = (size_t)(cols / 64) * (Kpad / 64);
    std::vector<unsigned char> rec(nrec * 15360);
    for (size_t r = 0; r < nrec; ++r) {
        unsigned char* dst = rec.data() + r * 15360;
        memcpy(dst, (const unsigned char*)wc.data() + r * 8192, 8192);
        memcpy(dst + 8192, ma.data() + r * 4096, 4096);
        memcpy(dst + 12288, mb.data() + r * 2048, 2048);
        memcpy(dst + 14336, (const unsigned char*)msc.data() + r * 1024, 1024);
    }
    return rec;
}

void fold_bn(const gdt_conv_desc& cd, const float* bias, const float* g, const float* b, const float* m, const float* v,
             std::vector<float>& scale, std::vector<float>& shift, bool& has_shift) {
    scale.assign(cd.cout, 1.f); shift.assign(cd.cout, 0.f);
    has_shift = bias != nullptr || g != nullptr;
    for (int c = 0; c < cd.cout; ++c) {
        float bs = bias ? bias[c] : 0.f;
        if (g) {
            const float s = g[c] / std::sqrt(v[c] + cd.bn_eps);
            scale[c] = s; shift[c] = b[c] + (bs - m[c]) * s;
        } else {
            shift[c] = bs;
        }
    }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

const char* gdt_last_error(void) { return g_last_error.c_str(); }
const char* gdt_version(void) { return "gandtr_hip 0.1 gfx950"; }

int gdt_net_create(gdt_net** net) {
    GDT_REQUIRE(net != nullptr, "net");
    *net = new gdt_net();
    std::vector<unsigned char> z(256, 0);
    (*net)->zeros_off = (*net)->blob_append(z.data(), z.size());
    return GDT_OK;
}

int gdt_net_set_precision(gdt_net* net, int mode) {
    GDT_REQUIRE(net && !net->finalized && net->ops.empty(), "precision must be chosen before the first op");
    GDT_REQUIRE(mode >= 0 && mode <= 3, "precision mode: 0 = f16, 1 = f16x3, 2 = f16c, 3 = f16ch (f16c + compensated head)");
    net->precision = mode == 3 ? 2 : mode;
    net->head_comp = mode == 3;
    return GDT_OK;
}

void gdt_net_destroy(gdt_net* net) {
    if (!net) return;
    if (net->dev_blob) (void)hipFree(net->dev_blob);
    for (hipEvent_t e : net->events) (void)hipEventDestroy(e);
    delete net;
}

int gdt_net_input(gdt_net* net, int channels, const int* perm, const float* scale, const float* shift, int* out_tensor) {
    GDT_REQUIRE(net && !net->finalized && out_tensor, "net");
    GDT_REQUIRE(net->input_op < 0, "only one external input per net");
    GDT_REQUIRE(channels >= 1 && channels <= 8, "input channels must be 1..8");
    Op o; o.kind = OP_INPUT; o.in_c = channels;
    for (int c = 0; c < 8; ++c) {
        o.perm[c] = (perm && c < channels) ? perm[c] : (c < channels ? c : 0);
        GDT_REQUIRE(o.perm[c] >= 0 && o.perm[c] < channels, "channel permutation out of range");
        o.scale[c] = (scale && c < channels) ? scale[c] : 1.f;
        o.shift[c] = (shift && c < channels) ? shift[c] : 0.f;
    }
    o.out = net->new_tensor(8, channels);
    net->input_op = (int)net->ops.size();
    net->ops.push_back(o);
    *out_tensor = o.out;
    return GDT_OK;
}

int gdt_net_conv(gdt_net* net, int in_tensor, const gdt_conv_desc* desc, const float* weight, const float* bias,
                 const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var,
                 int residual_tensor, int* out_tensor) {
    GDT_REQUIRE(net && !net->finalized && desc && weight && out_tensor, "net/desc/weight");
    GDT_REQUIRE(in_tensor >= 0 && in_tensor < (int)net->tensors.size(), "input tensor id");
    GDT_REQUIRE(residual_tensor < (int)net->tensors.size(), "residual tensor id");
    const gdt_conv_desc& cd = *desc;
    GDT_REQUIRE(cd.cin >= 1 && cd.cout >= 1 && cd.kh >= 1 && cd.kw >= 1 && cd.kh * cd.kw <= 64, "conv geometry");
    GDT_REQUIRE(cd.stride == 1 || cd.stride == 2, "stride must be 1 or 2");
    GDT_REQUIRE((bn_gamma && bn_beta && bn_mean && bn_var) || (!bn_gamma && !bn_beta && !bn_mean && !bn_var), "BN vectors");
    const int cin_pad = next_pow2(cd.cin);
    GDT_REQUIRE(net->tensors[in_tensor].C == cin_pad && net->tensors[in_tensor].Creal == cd.cin,
                "input tensor channel count does not match conv cin");
    if (cd.transposed) GDT_REQUIRE(cd.kh == 3 && cd.kw == 3 && cd.stride == 2 && cd.pad == 1 && !cd.pad_reflect,
                                   "only ConvTranspose2d(k3,s2,p1,op1) is supported");
    if (!cd.out_f32_nchw) GDT_REQUIRE(cd.cout % 8 == 0, "internal conv outputs need cout % 8 == 0");
    if (residual_tensor >= 0) GDT_REQUIRE(net->tensors[residual_tensor].C == cd.cout && !cd.out_f32_nchw, "residual channels");

    Op o; o.kind = OP_CONV; o.in = in_tensor; o.res = residual_tensor; o.cd = cd; o.cin_pad = cin_pad;
    o.rowsplit = cd.out_f32_nchw && !cd.transposed && cd.stride == 1 && cd.kw >= 3 && cd.cout <= 4 && cd.cout * cd.kw <= 32 &&
                 cd.kw == 2 * cd.pad + 1 && !cd.relu && !bn_gamma;
    const int gemm_cout = o.rowsplit ? cd.cout * cd.kw : cd.cout;
    if (o.rowsplit) o.rs_cout8 = (gemm_cout + 7) / 8 * 8;
    const int bn_tile = gdt_conv_bn(gemm_cout);
    o.cout_pad = (gemm_cout + bn_tile - 1) / bn_tile * bn_tile;

    std::vector<float> scale, shift; bool has_shift;
    fold_bn(cd, bias, bn_gamma, bn_beta, bn_mean, bn_var, scale, shift, has_shift);
    if (has_shift && o.rowsplit) {
        o.rs_bias_off = net->blob_append(shift.data(), shift.size() * sizeof(float));
        o.has_bias = true;     // applied by the combine kernel, not by the GEMM epilogue
    } else if (has_shift) {
        std::vector<float> bp(o.cout_pad, 0.f);
        std::copy(shift.begin(), shift.end(), bp.begin());
        o.bias_off = net->blob_append(bp.data(), bp.size() * sizeof(float));
        o.has_bias = true;
        if (!net->precision && cd.kh == 1 && cd.kw == 1 && cd.cin == 256 && o.cout_pad % 256 == 0) {
            // conv3x3_expand_rb.hip adds the expand conv's bias as one more k-step of its GEMM: per 32-channel block a weight fragment whose lane
            // (channel, fh = 0) holds { fp16(b), fp16(b - fp16(b)), 0 .. } (the pixel operand of that step is { 1, 1, 0 .. })
            std::vector<f16> bf((size_t)o.cout_pad / 32 * 512, (f16)0.f);
            for (int c = 0; c < o.cout_pad; ++c) {
                const f16 hi = (f16)bp[c];
                bf[((size_t)(c / 32) * 64 + (c & 31)) * 8] = hi;
                bf[((size_t)(c / 32) * 64 + (c & 31)) * 8 + 1] = (f16)(bp[c] - (float)hi);
            }
            o.bias_frag_off = net->blob_append(bf.data(), bf.size() * sizeof(f16));
            o.has_bias_frag = true;
        }
    }

    auto pack = [&](PackedPhase& ph, auto&& wget) {   // wget(cout, c, tap) -> float
        const int K = ph.ntaps * cin_pad;
        ph.Kpad = (K + 63) / 64 * 64;
        std::vector<f16> pk((size_t)o.cout_pad * ph.Kpad, (f16)0.f), pl;
        std::vector<float> wf;
        if (net->precision) pl.assign(pk.size(), (f16)0.f);
        if (net->precision == 2) wf.assign(pk.size(), 0.f);
        for (int co = 0; co < cd.cout; ++co)
            for (int t = 0; t < ph.ntaps; ++t)
                for (int c = 0; c < cd.cin; ++c) {
                    const size_t idx = (size_t)co * ph.Kpad + (size_t)t * cin_pad + c;
                    const float w = wget(co, c, t) * scale[co];
                    pk[idx] = (f16)w;
                    if (net->precision) pl[idx] = (f16)((w - (float)pk[idx]) * 2048.f);
                    if (net->precision == 2) wf[idx] = w;
                }
        ph.w_off = net->blob_append(pk.data(), pk.size() * sizeof(f16));
        if (net->precision) ph.w_lo_off = net->blob_append(pl.data(), pl.size() * sizeof(f16));
        if (net->precision == 2 && cin_pad % 64 == 0 && o.cout_pad % 128 == 0) {   // conv3x3_halo_c.hip
            std::vector<unsigned char> ma, mb; std::vector<unsigned> msc; std::vector<f16> wc;
            pack_mx(wf, o.cout_pad, ph.Kpad, ma, mb, msc, wc);
            ph.wc_off = net->blob_append(wc.data(), wc.size() * sizeof(f16));
            ph.wmx_a_off = net->blob_append(ma.data(), ma.size());
            ph.wmx_b_off = net->blob_append(mb.data(), mb.size());
            ph.wmx_s_off = net->blob_append(msc.data(), msc.size() * sizeof(unsigned));
            ph.has_mx = true;
            if (cd.kh == 3 && cd.kw == 3 && cd.stride == 1 && !cd.transposed && ph.ntaps == 9 && o.cout_pad % 256 == 0) {   // conv3x3_halo_c16.hip
                // one record per (64 output channels, 64 k-values): [fp16 step 0][fp16 step 1][16-byte parts][8-byte parts][scales] = 15 KB, so that
                // a wave's weight stream is ONE sequential region (conv3x3_halo_c16.hip)
                const std::vector<unsigned char> rec = pack_records16(wf, o.cout_pad, ph.Kpad);
                ph.w16_off = net->blob_append(rec.data(), rec.size());
                ph.has_mx16 = true;
            }
        }
        if (!net->precision && cin_pad % 64 == 0 && o.cout_pad % 32 == 0) {      // conv3x3_halo_rb.hip / conv_igemm_rb.hip
            // fragment order: lane = fh * 32 + fr holds cout = cb * 32 + fr, k = ks * 16 + fh * 8 + e
            const int nks = ph.Kpad / 16;
            std::vector<f16> pf(pk.size());
            for (int cb = 0; cb < o.cout_pad / 32; ++cb)
                for (int ks = 0; ks < nks; ++ks)
                    for (int ln = 0; ln < 64; ++ln) {
                        const f16* src = pk.data() + (size_t)(cb * 32 + (ln & 31)) * ph.Kpad + ks * 16 + (ln >> 5) * 8;
                        std::copy(src, src + 8, pf.data() + (((size_t)cb * nks + ks) * 64 + ln) * 8);
                    }
            ph.w_frag_off = net->blob_append(pf.data(), pf.size() * sizeof(f16));
            ph.has_frag = true;
        } else if (net->precision != 0 && cin_pad == 8 && 2 * cd.cin <= 8 && o.cout_pad == 64 && cd.cout == 64 && !cd.transposed) {
            // conv_stem.hip, f16c form: W1 slots of a tap = [w_hi (cin), w_hi * 2^-8 (cin), 0 ..], W2 = [w - w_hi (cin), 0 ..]; fragments as below
            const int nks = (ph.ntaps + 1) / 2;
            std::vector<f16> p1((size_t)o.cout_pad * ph.Kpad, (f16)0.f), p2(p1.size(), (f16)0.f);
            for (int co = 0; co < cd.cout; ++co)
                for (int t = 0; t < ph.ntaps; ++t)
                    for (int c = 0; c < cd.cin; ++c) {
                        const float w = wget(co, c, t) * scale[co];
                        const f16 wh = (f16)w;
                        p1[(size_t)co * ph.Kpad + (size_t)t * 8 + c] = wh;
                        p1[(size_t)co * ph.Kpad + (size_t)t * 8 + cd.cin + c] = (f16)((float)wh * (1.f / 256.f));
                        p2[(size_t)co * ph.Kpad + (size_t)t * 8 + c] = (f16)(w - (float)wh);
                    }
            std::vector<f16> f1((size_t)nks * 2 * 64 * 8, (f16)0.f), f2(f1.size(), (f16)0.f);
            for (int ks = 0; ks < nks; ++ks)
                for (int j = 0; j < 2; ++j)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int e = 0; e < 8; ++e) {
                            const int k = ks * 16 + (ln >> 5) * 8 + e;
                            if (k < ph.Kpad) {
                                f1[(((size_t)ks * 2 + j) * 64 + ln) * 8 + e] = p1[(size_t)(j * 32 + (ln & 31)) * ph.Kpad + k];
                                f2[(((size_t)ks * 2 + j) * 64 + ln) * 8 + e] = p2[(size_t)(j * 32 + (ln & 31)) * ph.Kpad + k];
                            }
                        }
            ph.w_frag_off = net->blob_append(f1.data(), f1.size() * sizeof(f16));
            ph.w_frag2_off = net->blob_append(f2.data(), f2.size() * sizeof(f16));
            ph.has_frag = true; ph.has_aug = true;
        } else if (!net->precision && cin_pad == 8 && o.cout_pad == 64 && cd.cout == 64 && !cd.transposed) {
            // conv_stem.hip: one k-step = two taps x 8 channels; fragments [ks][column block j][lane][8], zero past the last tap
            const int nks = (ph.ntaps + 1) / 2;
            std::vector<f16> pf((size_t)nks * 2 * 64 * 8, (f16)0.f);
            for (int ks = 0; ks < nks; ++ks)
                for (int j = 0; j < 2; ++j)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int e = 0; e < 8; ++e) {
                            const int k = ks * 16 + (ln >> 5) * 8 + e;
                            if (k < ph.Kpad) pf[(((size_t)ks * 2 + j) * 64 + ln) * 8 + e] = pk[(size_t)(j * 32 + (ln & 31)) * ph.Kpad + k];
                        }
            ph.w_frag_off = net->blob_append(pf.data(), pf.size() * sizeof(f16));
            ph.has_frag = true;
            const bool k7s2 = cd.kh == 7 && cd.kw == 7 && cd.stride == 2 && cd.pad == 3, k3s1 = cd.kh == 3 && cd.kw == 3 && cd.stride == 1 && cd.pad == 1;
            if ((k7s2 || k3s1) && !cd.pad_reflect && cd.cin <= 3) {
                // conv_stem_pair_kernel: a kernel row is HPR k-steps of four taps; k-step ks = ty * HPR + h covers taps tx = 4h .. 4h + 3 of row ty; lane (fh, fr)
                // element e = tap 4h + 2fh + (e >> 2), channel slot e & 3 (3 real channels; taps past the kernel do not exist: zero)
                const int KS = cd.kh, HPR = (KS + 3) / 4, NKS = KS * HPR;
                std::vector<f16> pp((size_t)NKS * 2 * 64 * 8, (f16)0.f);
                for (int ks = 0; ks < NKS; ++ks)
                    for (int j = 0; j < 2; ++j)
                        for (int ln = 0; ln < 64; ++ln)
                            for (int e = 0; e < 8; ++e) {
                                const int ty = ks / HPR, tx = 4 * (ks % HPR) + 2 * (ln >> 5) + (e >> 2), ch = e & 3;
                                if (tx < KS && ch < cd.cin) pp[(((size_t)ks * 2 + j) * 64 + ln) * 8 + e] = pk[(size_t)(j * 32 + (ln & 31)) * ph.Kpad + (ty * KS + tx) * 8 + ch];
                            }
                ph.w_pair_off = net->blob_append(pp.data(), pp.size() * sizeof(f16));
                ph.has_pair = true;
            }
        }
    };

    if (o.rowsplit) {
        // GEMM output channel co' = kx * cout + co, taps = kernel rows
        PackedPhase ph;
        ph.ntaps = cd.kh; ph.TW = 1; ph.dy0 = -cd.pad; ph.dys = 1; ph.dx0 = 0; ph.dxs = 0;
        {
            const int K = ph.ntaps * cin_pad;
            ph.Kpad = (K + 63) / 64 * 64;
            std::vector<f16> pk((size_t)o.cout_pad * ph.Kpad, (f16)0.f), pl;
            if (net->precision) pl.assign(pk.size(), (f16)0.f);
            for (int kx = 0; kx < cd.kw; ++kx)
                for (int co = 0; co < cd.cout; ++co)
                    for (int ky = 0; ky < cd.kh; ++ky)
                        for (int c = 0; c < cd.cin; ++c) {
                            const size_t idx = (size_t)(kx * cd.cout + co) * ph.Kpad + (size_t)ky * cin_pad + c;
                            const float w = weight[(((size_t)co * cd.cin + c) * cd.kh + ky) * cd.kw + kx];
                            pk[idx] = (f16)w;
                            if (net->precision) pl[idx] = (f16)((w - (float)pk[idx]) * 2048.f);
                        }
            ph.w_off = net->blob_append(pk.data(), pk.size() * sizeof(f16));
            if (net->precision) ph.w_lo_off = net->blob_append(pl.data(), pl.size() * sizeof(f16));
            static const bool head7_x3 = [] { const char* e = getenv("GDT_HEAD7_X3"); return !(e && atoi(e) == 0); }();      // 0: the f16x3 head stays on the generic GEMM + combine launch (A/B)
            if ((net->precision != 1 || head7_x3) && cin_pad == 64 && cd.kh == 7 && cd.kw == 7 && o.cout_pad == 32) {
                // conv_head7.hip keeps the whole matrix in registers: B fragment ks of lane (fh, fr) = column fr, k = ks*16 + fh*8 ..
                const int nks = ph.Kpad / 16;
                std::vector<f16> pf((size_t)nks * 64 * 8);
                for (int ks = 0; ks < nks; ++ks)
                    for (int ln = 0; ln < 64; ++ln) {
                        const f16* src = pk.data() + (size_t)(ln & 31) * ph.Kpad + ks * 16 + (ln >> 5) * 8;
                        std::copy(src, src + 8, pf.data() + ((size_t)ks * 64 + ln) * 8);
                    }
                ph.w_frag_off = net->blob_append(pf.data(), pf.size() * sizeof(f16));
                ph.has_frag = true;
                if (net->precision == 1) {                         // f16x3: the lo parts in the same fragment order (conv_head7.hip X3 form: ConvLaunch::w_frag2)
                    for (int ks = 0; ks < nks; ++ks)
                        for (int ln = 0; ln < 64; ++ln) {
                            const f16* src = pl.data() + (size_t)(ln & 31) * ph.Kpad + ks * 16 + (ln >> 5) * 8;
                            std::copy(src, src + 8, pf.data() + ((size_t)ks * 64 + ln) * 8);
                        }
                    ph.w_frag2_off = net->blob_append(pf.data(), pf.size() * sizeof(f16));
                }
                if (net->precision == 2 && net->head_comp) {       // "f16ch": block-scaled correction operands of the same [32][Kpad] matrix (conv_head7.hip, second pass)
                    std::vector<float> wf((size_t)o.cout_pad * ph.Kpad, 0.f);
                    for (int kx = 0; kx < cd.kw; ++kx)
                        for (int co = 0; co < cd.cout; ++co)
                            for (int ky = 0; ky < cd.kh; ++ky)
                                for (int c = 0; c < cd.cin; ++c)
                                    wf[(size_t)(kx * cd.cout + co) * ph.Kpad + (size_t)ky * cin_pad + c] = weight[(((size_t)co * cd.cin + c) * cd.kh + ky) * cd.kw + kx];
                    std::vector<unsigned char> ma, mb; std::vector<unsigned> msc; std::vector<f16> wc;
                    pack_mx(wf, o.cout_pad, ph.Kpad, ma, mb, msc, wc);
                    ph.wc_off = net->blob_append(wc.data(), wc.size() * sizeof(f16));
                    ph.wmx_a_off = net->blob_append(ma.data(), ma.size());
                    ph.wmx_b_off = net->blob_append(mb.data(), mb.size());
                    ph.wmx_s_off = net->blob_append(msc.data(), msc.size() * sizeof(unsigned));
                    ph.has_mx = true;
                }
            }
        }
        o.phases.push_back(ph);
    } else if (!cd.transposed) {
        PackedPhase ph;
        ph.ntaps = cd.kh * cd.kw; ph.TW = cd.kw; ph.dy0 = -cd.pad; ph.dys = 1; ph.dx0 = -cd.pad; ph.dxs = 1;
        const int khw = cd.kh * cd.kw;
        pack(ph, [&](int co, int c, int t) { return weight[((size_t)co * cd.cin + c) * khw + t]; });
        o.phases.push_back(ph);
        if (net->precision != 0 && cd.stride == 2 && cd.kh == 3 && cd.kw == 3 && cd.pad == 1 && !cd.pad_reflect && !cd.out_f32_nchw &&
            residual_tensor < 0 && (cin_pad == 64 || cin_pad == 128) && cd.cin == cin_pad) {
            // shift form: K index = shift * 4cin + parity * cin + c, shift = (dy+1)*2 + (dx+1) with dy, dx in {-1, 0}, parity = py*2 + px of the
            // input pixel (2R + py, 2C + px); kernel row ky = 0 for (dy -1, py 1), 1 for (0, 0), 2 for (0, 1), none for (-1, 0); columns alike
            PackedPhase& sp = o.s2;
            sp.ntaps = 4; sp.TW = 2; sp.dy0 = -1; sp.dys = 1; sp.dx0 = -1; sp.dxs = 1; sp.Kpad = 16 * cin_pad;
            o.s2_cout_pad = (cd.cout + 127) / 128 * 128;             // (128-column tiles for Cout <= 128: no padding columns to multiply)
            std::vector<float> wf((size_t)o.s2_cout_pad * sp.Kpad, 0.f);
            auto tap_of = [](int shift, int par) { return shift == 0 ? (par == 1 ? 0 : -1) : (par == 0 ? 1 : 2); };
            for (int co = 0; co < cd.cout; ++co)
                for (int t = 0; t < 4; ++t)
                    for (int par = 0; par < 4; ++par) {
                        const int ky = tap_of(t >> 1, par >> 1), kx = tap_of(t & 1, par & 1);
                        if (ky < 0 || kx < 0) continue;
                        for (int c = 0; c < cd.cin; ++c)
                            wf[(size_t)co * sp.Kpad + (size_t)t * 4 * cin_pad + (size_t)par * cin_pad + c] =
                                weight[((size_t)co * cd.cin + c) * 9 + ky * 3 + kx] * scale[co];
                    }
            if (net->precision == 1) {          // f16x3: the same matrix split into hi / lo, row-major (conv3x3_halo_x3.hip FORM 2)
                std::vector<f16> pk(wf.size()), pl(wf.size());
                for (size_t i = 0; i < wf.size(); ++i) { pk[i] = (f16)wf[i]; pl[i] = (f16)((wf[i] - (float)pk[i]) * 2048.f); }
                sp.w_off = net->blob_append(pk.data(), pk.size() * sizeof(f16));
                sp.w_lo_off = net->blob_append(pl.data(), pl.size() * sizeof(f16));
            } else {
            std::vector<unsigned char> ma, mb; std::vector<unsigned> msc; std::vector<f16> wc;
            pack_mx(wf, o.s2_cout_pad, sp.Kpad, ma, mb, msc, wc);
            sp.wc_off = net->blob_append(wc.data(), wc.size() * sizeof(f16));
            sp.wmx_a_off = net->blob_append(ma.data(), ma.size());
            sp.wmx_b_off = net->blob_append(mb.data(), mb.size());
            sp.wmx_s_off = net->blob_append(msc.data(), msc.size() * sizeof(unsigned));
            sp.has_mx = true;
            }
            if (has_shift) {
                std::vector<float> bp(o.s2_cout_pad, 0.f);
                std::copy(shift.begin(), shift.end(), bp.begin());
                o.s2_bias_off = net->blob_append(bp.data(), bp.size() * sizeof(float));
            }
            o.has_s2 = true;
        }
    } else {
        // o = 2i - 1 + k.  Even outputs (parity 0): k = 1, i = y.  Odd outputs: k = 0 (i = y + 1) and k = 2 (i = y).
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                PackedPhase ph;
                const int th = py ? 2 : 1, tw = px ? 2 : 1;
                ph.ntaps = th * tw; ph.TW = tw;
                ph.dy0 = py ? 1 : 0; ph.dys = -1; ph.dx0 = px ? 1 : 0; ph.dxs = -1;
                ph.ooy = py; ph.oox = px;
                pack(ph, [&](int co, int c, int t) {
                    const int ty = t / tw, tx = t % tw;
                    const int ky = py ? (ty == 0 ? 0 : 2) : 1, kx = px ? (tx == 0 ? 0 : 2) : 1;
                    return weight[(((size_t)c * cd.cout + co) * 3 + ky) * 3 + kx];
                });
                o.phases.push_back(ph);
            }
        if (net->precision == 1 && cd.cout == 64 && cin_pad % 32 == 0 && cd.cin == cin_pad && residual_tensor < 0) {
            for (int py = 0; py < 2; ++py) {
                PackedPhase pp;
                const int th = py ? 2 : 1, tw = 2;
                pp.ntaps = th * tw; pp.TW = tw; pp.dy0 = py ? 1 : 0; pp.dys = -1; pp.dx0 = 1; pp.dxs = -1;
                pp.ooy = py; pp.oox = 0; pp.ooy2 = py; pp.oox2 = 1;
                pp.Kpad = pp.ntaps * cin_pad;
                std::vector<f16> pk((size_t)128 * pp.Kpad, (f16)0.f), pl(pk.size(), (f16)0.f);
                for (int px = 0; px < 2; ++px)
                    for (int t = 0; t < pp.ntaps; ++t) {
                        const int dy = pp.dy0 + (t / tw) * pp.dys, dx = pp.dx0 + (t % tw) * pp.dxs;
                        const int ky = py ? (dy ? 0 : 2) : (dy ? -1 : 1), kx = px ? (dx ? 0 : 2) : (dx ? -1 : 1);
                        if (ky < 0 || kx < 0) continue;                       // (this phase does not see this shift: a zero block)
                        for (int co = 0; co < cd.cout; ++co)
                            for (int c = 0; c < cd.cin; ++c) {
                                const float w = weight[(((size_t)c * cd.cout + co) * 3 + ky) * 3 + kx] * scale[co];
                                const size_t idx = (size_t)(px * 64 + co) * pp.Kpad + (size_t)t * cin_pad + c;
                                pk[idx] = (f16)w;
                                pl[idx] = (f16)((w - (float)pk[idx]) * 2048.f);
                            }
                    }
                pp.w_off = net->blob_append(pk.data(), pk.size() * sizeof(f16));
                pp.w_lo_off = net->blob_append(pl.data(), pl.size() * sizeof(f16));
                o.pairs.push_back(pp);
            }
            if (has_shift) {
                std::vector<float> b2(128);
                for (int i = 0; i < 128; ++i) b2[i] = shift[i & 63];
                o.pair_bias_off = net->blob_append(b2.data(), b2.size() * sizeof(float));
            }
            o.has_pairs = true;
        }
        if (net->precision != 1 && cin_pad % 64 == 0 && (4 * cd.cout) % 256 == 0 && 256 % cd.cout == 0 && cd.cout >= 64 && residual_tensor < 0) {
            // fused form: GEMM column -> (phase py * 2 + px, co) by gdt_ctf_column(), k = (dy * 2 + dx) * cin + c over the 2x2 input
            // shifts; a (shift, phase) pair that does not occur is a zero block (16 blocks, 9 non-zero) the kernel skips
            PackedPhase& cf = o.ctf;
            cf.ntaps = 4; cf.TW = 2; cf.dy0 = 0; cf.dys = 1; cf.dx0 = 0; cf.dxs = 1; cf.Kpad = 4 * cin_pad;
            const int ncol = 4 * cd.cout, nks = cf.Kpad / 16;
            std::vector<f16> pk((size_t)ncol * cf.Kpad, (f16)0.f);
            std::vector<float> wf;
            if (net->precision == 2) wf.assign(pk.size(), 0.f);
            for (int col = 0; col < ncol; ++col) {
                int phase, co;
                if (net->precision == 2) gdt_ctc_column(col, phase, co); else gdt_ctf_column(col, cd.cout, phase, co);
                const int py = phase >> 1, px = phase & 1;
                for (int dy = 0; dy < 2; ++dy)
                    for (int dx = 0; dx < 2; ++dx) {
                        const int ky = py ? (dy ? 0 : 2) : (dy ? -1 : 1), kx = px ? (dx ? 0 : 2) : (dx ? -1 : 1);
                        if (ky < 0 || kx < 0) continue;
                        for (int c = 0; c < cd.cin; ++c) {
                            const float w = weight[(((size_t)c * cd.cout + co) * 3 + ky) * 3 + kx] * scale[co];
                            pk[(size_t)col * cf.Kpad + (size_t)(dy * 2 + dx) * cin_pad + c] = (f16)w;
                            if (net->precision == 2) wf[(size_t)col * cf.Kpad + (size_t)(dy * 2 + dx) * cin_pad + c] = w;
                        }
                    }
            }
            if (net->precision == 2) {          // conv3x3_halo_c.hip, transposed form
                std::vector<unsigned char> ma, mb; std::vector<unsigned> msc; std::vector<f16> wc;
                pack_mx(wf, ncol, cf.Kpad, ma, mb, msc, wc);
                cf.wc_off = net->blob_append(wc.data(), wc.size() * sizeof(f16));
                cf.wmx_a_off = net->blob_append(ma.data(), ma.size());
                cf.wmx_b_off = net->blob_append(mb.data(), mb.size());
                cf.wmx_s_off = net->blob_append(msc.data(), msc.size() * sizeof(unsigned));
                cf.has_mx = true;
            }
            std::vector<f16> pf(net->precision ? 0 : pk.size());
            for (int cb = 0; cb < (net->precision ? 0 : ncol / 32); ++cb)
                for (int ks = 0; ks < nks; ++ks)
                    for (int ln = 0; ln < 64; ++ln) {
                        const f16* src = pk.data() + (size_t)(cb * 32 + (ln & 31)) * cf.Kpad + ks * 16 + (ln >> 5) * 8;
                        std::copy(src, src + 8, pf.data() + (((size_t)cb * nks + ks) * 64 + ln) * 8);
                    }
            if (!net->precision) { cf.w_frag_off = net->blob_append(pf.data(), pf.size() * sizeof(f16)); cf.has_frag = true; }
            if (has_shift) {
                std::vector<float> b4(ncol);
                for (int i = 0; i < ncol; ++i) { int ph, co; if (net->precision == 2) gdt_ctc_column(i, ph, co); else gdt_ctf_column(i, cd.cout, ph, co); b4[i] = shift[co]; }
                o.ctf_bias_off = net->blob_append(b4.data(), b4.size() * sizeof(float));
            }
            o.has_ctf = true;
        }
    }
    if (cd.out_f32_nchw) {
        o.slot = (int)net->out_ops.size();
        net->out_ops.push_back((int)net->ops.size());
        *out_tensor = o.slot;
    } else {
        o.out = net->new_tensor(cd.cout, cd.cout);
        *out_tensor = o.out;
    }
    net->ops.push_back(std::move(o));
    return GDT_OK;
}

int gdt_net_instance_norm(gdt_net* net, int in_tensor, float eps, int relu, int residual_tensor, int* out_tensor) {
    GDT_REQUIRE(net && !net->finalized && out_tensor, "net");
    GDT_REQUIRE(in_tensor >= 0 && in_tensor < (int)net->tensors.size() && residual_tensor < (int)net->tensors.size(), "tensor id");
    const int C = net->tensors[in_tensor].C;
    GDT_REQUIRE((C & (C - 1)) == 0 && C >= 8 && C <= 2048, "InstanceNorm needs a power-of-two channel count in [8, 2048]");
    if (residual_tensor >= 0) GDT_REQUIRE(net->tensors[residual_tensor].C == C, "residual channels");
    Op o; o.kind = OP_INORM; o.in = in_tensor; o.res = residual_tensor; o.eps = eps; o.relu = relu;
    o.out = net->new_tensor(C, net->tensors[in_tensor].Creal);
    for (size_t k = 0; k < net->ops.size(); ++k)
        if (net->ops[k].kind == OP_CONV && net->ops[k].out == in_tensor && net->ops[k].stats_for < 0) {
            net->ops[k].stats_for = (int)net->ops.size();
            o.stats_from = (int)k;
        }
    net->ops.push_back(o);
    *out_tensor = o.out;
    return GDT_OK;
}

int gdt_net_maxpool(gdt_net* net, int in_tensor, int kernel, int stride, int pad, int* out_tensor) {
    GDT_REQUIRE(net && !net->finalized && out_tensor, "net");
    GDT_REQUIRE(in_tensor >= 0 && in_tensor < (int)net->tensors.size(), "tensor id");
    GDT_REQUIRE(kernel >= 1 && stride >= 1 && pad >= 0 && pad * 2 <= kernel, "maxpool geometry");
    Op o; o.kind = OP_MAXPOOL; o.in = in_tensor; o.k = kernel; o.s = stride; o.p = pad;
    o.out = net->new_tensor(net->tensors[in_tensor].C, net->tensors[in_tensor].Creal);
    net->ops.push_back(o);
    *out_tensor = o.out;
    return GDT_OK;
}

int gdt_net_gem_l2n(gdt_net* net, int in_tensor, float p, float eps_gem, float eps_l2, int* out_slot) {
    GDT_REQUIRE(net && !net->finalized && out_slot, "net");
    GDT_REQUIRE(in_tensor >= 0 && in_tensor < (int)net->tensors.size(), "tensor id");
    GDT_REQUIRE(net->tensors[in_tensor].C % 64 == 0, "GeM needs channels % 64 == 0");
    GDT_REQUIRE(p > 0.f, "GeM exponent must be positive");
    Op o; o.kind = OP_GEM; o.in = in_tensor; o.gem_p = p; o.eps_gem = eps_gem; o.eps_l2 = eps_l2;
    o.slot = (int)net->out_ops.size();
    net->out_ops.push_back((int)net->ops.size());
    net->ops.push_back(o);
    *out_slot = o.slot;
    return GDT_OK;
}

int gdt_net_output_nchw(gdt_net* net, int in_tensor, const float* bias, int* out_slot) {
    GDT_REQUIRE(net && !net->finalized && out_slot, "net");
    GDT_REQUIRE(in_tensor >= 0 && in_tensor < (int)net->tensors.size(), "tensor id");
    Op o; o.kind = OP_OUT_NCHW; o.in = in_tensor;
    if (bias) { o.tap_bias_off = net->blob_append(bias, net->tensors[in_tensor].C * sizeof(float)); o.tap_has_bias = true; }
    o.slot = (int)net->out_ops.size();
    net->out_ops.push_back((int)net->ops.size());
    net->ops.push_back(o);
    *out_slot = o.slot;
    return GDT_OK;
}

int gdt_net_hed_head(gdt_net* net, const int* feature_tensors, const float* const* score_w, const float* score_b,
                     const float* fusion_w, float fusion_b, int sigmoid, int* out_slot) {
    GDT_REQUIRE(net && !net->finalized && feature_tensors && score_w && score_b && fusion_w && out_slot, "net/args");
    Op o; o.kind = OP_HED; o.sigmoid = sigmoid; o.fusion_b = fusion_b;
    for (int k = 0; k < 5; ++k) {
        GDT_REQUIRE(feature_tensors[k] >= 0 && feature_tensors[k] < (int)net->tensors.size(), "tensor id");
        o.feats[k] = feature_tensors[k];
        o.score_w_off[k] = net->blob_append(score_w[k], net->tensors[o.feats[k]].C * sizeof(float));
        o.score_b[k] = score_b[k]; o.fusion_w[k] = fusion_w[k];
    }
    o.slot = (int)net->out_ops.size();
    net->out_ops.push_back((int)net->ops.size());
    net->ops.push_back(o);
    *out_slot = o.slot;
    return GDT_OK;
}

// fp16 mode: for every 1x1 conv c (stride 1, ReLU) whose residual is the output of a 1x1 projection conv ds (no ReLU, stride 1 or 2, no other consumer),
// append the fragment-ordered K-concatenation [W_c | W_ds] and the summed bias to the weight blob (see Op::kcat_ds); whether a forward uses it is the
// planner's decision per geometry
static void build_kcat_weights(gdt_net* net) {
    if (net->kcat_built) return;
    net->kcat_built = true;
    auto& ops = net->ops;
    std::vector<int> consumers(net->tensors.size(), 0);
    for (const Op& o : ops) {
        if (o.in >= 0) ++consumers[o.in];
        if (o.res >= 0) ++consumers[o.res];
        if (o.kind == OP_HED) for (int k = 0; k < 5; ++k) ++consumers[o.feats[k]];
    }
    auto plain1x1 = [](const Op& o) {
        return o.kind == OP_CONV && !o.cd.transposed && !o.cd.out_f32_nchw && !o.rowsplit && o.stats_for < 0 && o.phases.size() == 1 && o.phases[0].has_frag &&
               o.has_bias && o.cd.kh == 1 && o.cd.kw == 1 && o.cd.pad == 0 && o.cin_pad % 64 == 0 && o.phases[0].Kpad == o.cin_pad && o.out >= 0;
    };
    for (size_t i = 0; i < ops.size(); ++i) {
        Op& c = ops[i];
        if (!plain1x1(c) || c.cd.stride != 1 || !c.cd.relu || c.res < 0 || c.cout_pad % 256 != 0) continue;
        int ids = -1;
        for (size_t j = 0; j < i; ++j) if (ops[j].out == c.res) ids = (int)j;
        if (ids < 0) continue;
        const Op& ds = ops[ids];
        if (!plain1x1(ds) || ds.cd.relu || ds.res >= 0 || ds.cd.stride < 1 || ds.cd.stride > 2 || ds.cout_pad != c.cout_pad || ds.cd.cout != c.cd.cout || consumers[ds.out] != 1) continue;
        const int K1 = c.cin_pad, K2 = ds.cin_pad, K = K1 + K2, nks = K / 16, cp = c.cout_pad;
        if ((K / 64) % 2 != 0) continue;
        const f16* w1 = (const f16*)(net->host_blob.data() + c.phases[0].w_off);
        const f16* w2 = (const f16*)(net->host_blob.data() + ds.phases[0].w_off);
        std::vector<f16> pf((size_t)cp * K);
        for (int cb = 0; cb < cp / 32; ++cb)
            for (int ks = 0; ks < nks; ++ks)
                for (int ln = 0; ln < 64; ++ln) {
                    const int co = cb * 32 + (ln & 31), k0 = ks * 16 + (ln >> 5) * 8;         // (a group of 8 k never straddles the two matrices: K1 % 64 == 0)
                    const f16* src = k0 < K1 ? w1 + (size_t)co * K1 + k0 : w2 + (size_t)co * K2 + (k0 - K1);
                    std::copy(src, src + 8, pf.data() + (((size_t)cb * nks + ks) * 64 + ln) * 8);
                }
        std::vector<float> bsum(cp);
        const float* b1 = (const float*)(net->host_blob.data() + c.bias_off);
        const float* b2 = (const float*)(net->host_blob.data() + ds.bias_off);
        for (int k = 0; k < cp; ++k) bsum[k] = b1[k] + b2[k];
        c.kcat_frag_off = net->blob_append(pf.data(), pf.size() * sizeof(f16));           // (invalidates w1 / w2 / b1 / b2: not used below)
        c.kcat_bias_off = net->blob_append(bsum.data(), bsum.size() * sizeof(float));
        c.kcat_ds = ids;
    }
}

int gdt_net_finalize(gdt_net* net) {
    GDT_REQUIRE(net && !net->finalized, "net");
    GDT_REQUIRE(net->input_op == 0, "the first op must be gdt_net_input");
    if (!net->precision) build_kcat_weights(net);
    const size_t bytes = align_up(net->host_blob.size());
    net->host_blob.resize(bytes);
    GDT_CHECK_HIP(hipMalloc((void**)&net->dev_blob, bytes));
    GDT_CHECK_HIP(hipMemcpy(net->dev_blob, net->host_blob.data(), bytes, hipMemcpyHostToDevice));
    std::vector<unsigned char>().swap(net->host_blob);
    net->finalized = true;
    return GDT_OK;
}

int gdt_net_num_outputs(gdt_net* net) { return net ? (int)net->out_ops.size() : 0; }

int gdt_net_output_shape(gdt_net* net, int slot, int n, int rh, int rw, int* dims, int* ndim) {
    GDT_REQUIRE(net && dims && ndim && slot >= 0 && slot < (int)net->out_ops.size(), "slot");
    Plan plan;
    int rc = make_plan(net, n, rh, rw, plan);
    if (rc != GDT_OK) return rc;
    const Op& o = net->ops[net->out_ops[slot]];
    const Tensor& ti = net->tensors[o.kind == OP_HED ? 0 : o.in];
    switch (o.kind) {
        case OP_CONV:
            dims[0] = n; dims[1] = o.cd.cout; dims[2] = conv_out_dim(o.cd, ti.H, o.cd.kh); dims[3] = conv_out_dim(o.cd, ti.W, o.cd.kw);
            *ndim = 4; break;
        case OP_GEM: dims[0] = n; dims[1] = ti.C; *ndim = 2; break;
        case OP_OUT_NCHW: dims[0] = n; dims[1] = ti.C; dims[2] = ti.H; dims[3] = ti.W; *ndim = 4; break;
        case OP_HED: dims[0] = n; dims[1] = 1; dims[2] = rh; dims[3] = rw; *ndim = 4; break;
        default: GDT_REQUIRE(false, "not an output op");
    }
    return GDT_OK;
}

int gdt_net_workspace_bytes(gdt_net* net, int n, int rh, int rw, size_t* bytes) {
    GDT_REQUIRE(net && bytes && n >= 1 && rh >= 1 && rw >= 1, "geometry");
    Plan plan, direct;
    int rc = make_plan(net, n, rh, rw, plan);
    if (rc != GDT_OK) return rc;
    rc = make_plan(net, n, rh, rw, direct, true);          // a call that does not resize may take the direct-stem plan: another layout, either may be the larger
    if (rc != GDT_OK) return rc;
    *bytes = std::max(plan.peak, direct.peak) + ALIGN;
    return GDT_OK;
}

static double op_flops(const gdt_net* net, const Op& o, int n, int rh, int rw) {
    if (o.kind == OP_CONV) {
        const Tensor& ti = net->tensors[o.in];
        if (o.cd.transposed)   // every input pixel meets every kernel tap once
            return 2.0 * n * ti.H * ti.W * (double)o.cd.cin * o.cd.cout * o.cd.kh * o.cd.kw;
        return 2.0 * n * (double)conv_out_dim(o.cd, ti.H, o.cd.kh) * conv_out_dim(o.cd, ti.W, o.cd.kw) * o.cd.cin * o.cd.cout *
               o.cd.kh * o.cd.kw;
    }
    if (o.kind == OP_HED) {
        double f = 2.0 * n * rh * rw * 5;
        for (int k = 0; k < 5; ++k) { const Tensor& tf = net->tensors[o.feats[k]]; f += 2.0 * n * tf.H * tf.W * tf.C; }
        return f;
    }
    return 0.0;
}

// Algorithmic HBM bytes of a conv op as its own launch: the input tensor once (a strided 1x1 conv touches only the pixels it samples), the
// output once, the residual once, the fp16 weights once ("each conv reads its input and writes its output once").  Fused
// launches subtract the tensors that never exist (see the forward).
static double op_bytes(const gdt_net* net, const Op& o, int n) {
    if (o.kind != OP_CONV) return 0.0;
    const double es = (double)net->esize();
    const Tensor& ti = net->tensors[o.in];
    // (output geometry from the conv itself: a conv that writes a caller-facing fp32 NCHW slot has no internal output tensor)
    const double oh = conv_out_dim(o.cd, ti.H, o.cd.kh), ow = conv_out_dim(o.cd, ti.W, o.cd.kw);
    const double out_b = (double)n * oh * ow * o.cd.cout * (o.cd.out_f32_nchw ? 4.0 : es);
    double in_px = (double)n * ti.H * ti.W;
    if (!o.cd.transposed && o.cd.kh == 1 && o.cd.kw == 1 && o.cd.stride > 1) in_px = (double)n * oh * ow;
    double b = in_px * ti.C * es + out_b;
    if (o.res >= 0) b += out_b;
    b += (double)o.cd.cin * o.cd.cout * o.cd.kh * o.cd.kw * sizeof(f16);
    return b;
}

// What the planner decides for a geometry, as counts (host logic only: no device call) -- so that the fusion decisions are testable without a GPU.
// counts[0] launches of conv ops (a fused launch counts once), [1] whole Bottlenecks in one launch (conv_bneck.hip), [2] 3x3 + expand launches (conv3x3_expand_rb.hip),
// [3] of those with the next block's reduce conv chained in, [4] projection shortcuts folded into their expand conv (K-concatenated 1x1), [5] InstanceNorms applied by
// their consumer's staging, [6] max-pools written by their producer, [7] 1 if the stem reads the caller's image itself (calls that do not resize),
// [8] transposed convs as one fused-phase launch, [9] stride-2 convs as the shift form
int gdt_net_plan_summary(gdt_net* net, int n, int rh, int rw, int resize, int* counts, int n_counts) {
    GDT_REQUIRE(net && counts && n_counts >= 10 && n >= 1 && rh >= 1 && rw >= 1, "plan summary arguments");
    if (!net->finalized && !net->precision) build_kcat_weights(net);       // (what finalize would add: the K-concatenated shortcut weights the planner may choose)
    Plan plan;
    int rc = make_plan(net, n, rh, rw, plan, resize == 0);
    if (rc != GDT_OK) return rc;
    for (int k = 0; k < n_counts; ++k) counts[k] = 0;
    for (size_t i = 0; i < plan.steps.size(); ++i) {
        const Step& st = plan.steps[i];
        const Op& o = net->ops[i];
        if (o.kind == OP_CONV && !st.skip) ++counts[0];
        if (o.kind == OP_CONV) { counts[1] += st.bneck; counts[2] += st.xexp; counts[3] += st.xchain >= 0; counts[4] += st.kcat; counts[8] += st.ctf; counts[9] += st.s2; }
        if (o.kind == OP_INORM) counts[5] += st.norm_into >= 0;
        if (o.kind == OP_MAXPOOL) counts[6] += st.skip;
        if (o.kind == OP_INPUT) counts[7] += st.direct;
    }
    return GDT_OK;
}

int gdt_net_flops(gdt_net* net, int n, int rh, int rw, double* flops) {
    GDT_REQUIRE(net && flops, "net");
    Plan plan;
    int rc = make_plan(net, n, rh, rw, plan);
    if (rc != GDT_OK) return rc;
    double f = 0.0;
    for (const Op& o : net->ops) f += op_flops(net, o, n, rh, rw);
    *flops = f;
    return GDT_OK;
}

int gdt_net_set_profiling(gdt_net* net, int enable) {
    GDT_REQUIRE(net && net->finalized, "net must be finalized");
    if (enable && net->events.empty()) {
        net->events.resize(net->ops.size() * 2);
        for (auto& e : net->events) GDT_CHECK_HIP(hipEventCreate(&e));
    }
    net->profiling = enable != 0;
    return GDT_OK;
}

int gdt_net_profile_read(gdt_net* net, int max_ops, int* n_ops, int* kinds, int* tile_n, double* ms, double* flops) {
    GDT_REQUIRE(net && n_ops && kinds && tile_n && ms && flops, "profile buffers");
    GDT_REQUIRE(!net->events.empty() && net->last_flops.size() == net->ops.size(), "no profiled forward has run");
    const int n = (int)net->ops.size();
    GDT_REQUIRE(max_ops >= n, "profile buffers too small");
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        GDT_CHECK_HIP(hipEventSynchronize(net->events[2 * i + 1]));
        GDT_CHECK_HIP(hipEventElapsedTime(&t, net->events[2 * i], net->events[2 * i + 1]));
        kinds[i] = (int)net->ops[i].kind;
        tile_n[i] = net->last_variant[i];
        ms[i] = t;
        flops[i] = net->last_flops[i];
    }
    *n_ops = n;
    return GDT_OK;
}

int gdt_net_num_ops(gdt_net* net) { return net ? (int)net->ops.size() : 0; }

int gdt_net_profile_read_bytes(gdt_net* net, int max_ops, int* n_ops, double* bytes) {
    GDT_REQUIRE(net && bytes && n_ops, "profile buffers");
    GDT_REQUIRE(net->last_bytes.size() == net->ops.size(), "no profiled forward has run");
    GDT_REQUIRE(max_ops >= (int)net->ops.size(), "profile buffers too small");
    for (size_t i = 0; i < net->ops.size(); ++i) bytes[i] = net->last_bytes[i];
    *n_ops = (int)net->ops.size();
    return GDT_OK;
}

}  // extern "C"

namespace {

// one geometry of a forward: its input, outputs, workspace and the plan made for it (a snapshot of the planned tensor table: make_plan works on net->tensors)
struct LevelCtx {
    const float* x; int n, h, w, rh, rw; float rscale;
    void* const* outputs; char* ws;
    Plan plan; std::vector<Tensor> T;
    float group_factor = 1.f;                    // the planner hint this geometry was planned with (gdt_net_set_group_factor)
};
enum { DEFER_NONE = 0, DEFER_CONV = 1, DEFER_BNECK = 2 };
// a launch a step WOULD make, handed back instead of issued: the lock-step driver (several geometries per forward) joins the levels' launches of one op
struct Deferred {
    int kind = DEFER_NONE;
    bool kcat = false;
    ConvLaunch d;                                // DEFER_CONV: the descriptor gdt_launch_conv / gdt_launch_conv_1x1_rb (kcat) would get
    const f16* bx = nullptr; f16* by = nullptr; int bn = 0, bh = 0, bw = 0;      // DEFER_BNECK: block input / output and geometry (the weights are the op's)
};

// One op of the graph on one geometry.  `defer` != null: launches that can share a launch with the other geometries' are handed back (kind != DEFER_NONE) instead
// of issued.  `book`: per-op profile bookkeeping of fused launches (once per forward).
int exec_step(gdt_net* net, LevelCtx& c, const Step& stp, hipStream_t st, Deferred* defer, bool book) {
    const int n = c.n, h = c.h, w = c.w, rh = c.rh, rw = c.rw;
    const float rscale = c.rscale;
    const float* x = c.x;
    void* const* outputs = c.outputs;
    char* ws = c.ws;
    auto& T = c.T;
    const Plan& plan = c.plan;
    auto tptr = [&](int t) { return (f16*)(ws + T[t].off); };      // element type is fp16 or fp32 (net->precision)
    const int f32 = net->precision ? 1 : 0;      // activation element type handed to the helper kernels: fp32 in both split modes
    const f16* zeros = (const f16*)(net->dev_blob + net->zeros_off);
    const Op& o = net->ops[stp.op];
    int rc = GDT_OK;
    (void)h; (void)w; (void)rscale; (void)x;
        switch (o.kind) {
            case OP_INPUT: {
                const int resize = (rh != h || rw != w) ? 1 : 0;
                if (stp.direct) break;                       // the stem conv reads x itself (planned only for calls that do not resize)
                rc = gdt_k_pack_input(x, tptr(o.out), stp.aug ? 2 : f32, n, o.in_c, h, w, rh, rw, rscale, resize, o.perm, o.scale, o.shift, st);
                break;
            }
            case OP_CONV: {
                const Tensor& ti = T[o.in];
                if (stp.skip) break;           // second / third conv of a fused Bottleneck
                if (stp.bneck) {               // the whole Bottleneck in one launch (conv_bneck.hip)
                    const bool dsf = stp.bneck_ds >= 0;
                    const Op &oa = net->ops[stp.bneck_a], &ob = net->ops[stp.op + (dsf ? 2 : 1)], &oc = net->ops[stp.op + (dsf ? 3 : 2)];
                    const Op* od = dsf ? &net->ops[stp.bneck_ds] : nullptr;
                    if (defer) { defer->kind = DEFER_BNECK; defer->bx = tptr(oa.in); defer->by = tptr(oc.out); defer->bn = n; defer->bh = T[oa.in].H; defer->bw = T[oa.in].W; }
                    else
                    rc = gdt_launch_bneck(tptr(oa.in), tptr(oc.out), (const f16*)(net->dev_blob + oa.phases[0].w_frag_off),
                                          (const f16*)(net->dev_blob + ob.phases[0].w_frag_off), (const f16*)(net->dev_blob + oc.phases[0].w_frag_off),
                                          (const float*)(net->dev_blob + oa.bias_off), (const float*)(net->dev_blob + ob.bias_off),
                                          (const float*)(net->dev_blob + oc.bias_off),
                                          od ? (const f16*)(net->dev_blob + od->phases[0].w_frag_off) : nullptr, od ? (const float*)(net->dev_blob + od->bias_off) : nullptr,
                                          oa.cd.cin, oc.cd.cout, oa.cd.cout, n, T[oa.in].H, T[oa.in].W, st);
                    if (net->profiling && book) {      // the block's FLOPs and time are booked on its first conv
                        net->last_variant[stp.op] = 935000 + oc.cd.cout + (dsf ? 1 : 0);
                        for (int k = 1; k <= (dsf ? 3 : 2); ++k) { net->last_flops[stp.op] += net->last_flops[stp.op + k]; net->last_flops[stp.op + k] = 0.0; }
                        // bytes: the block-boundary tensors (x once -- it is also the residual --, y once) and every weight matrix once
                        const double es = (double)net->esize();
                        double by = (double)n * T[oa.in].H * T[oa.in].W * T[oa.in].C * es + (double)n * T[oc.out].H * T[oc.out].W * T[oc.out].C * es;
                        for (const Op* w : {&oa, &ob, &oc, od}) if (w) by += (double)w->cd.cin * w->cd.cout * w->cd.kh * w->cd.kw * sizeof(f16);
                        for (int k = 0; k <= (dsf ? 3 : 2); ++k) net->last_bytes[stp.op + k] = 0.0;
                        net->last_bytes[stp.op] = by;
                    }
                    break;
                }
                if (stp.direct) {              // ResNet stem from the fp32 NCHW image (conv_stem.hip, pair-word form), with the max-pool behind it when planned so
                    const Op& oi = net->ops[0];
                    ConvLaunch d{};
                    conv_geometry(net, o, o.phases[0], n, ti, d);
                    d.zeros = zeros;
                    d.w_frag = (const f16*)(net->dev_blob + o.phases[0].w_pair_off);
                    d.bias = o.has_bias ? (const float*)(net->dev_blob + o.bias_off) : nullptr;
                    if (stp.pool_into >= 0) {
                        const Tensor& tp = T[net->ops[stp.pool_into].out];
                        d.out = tptr(net->ops[stp.pool_into].out);
                        rc = gdt_launch_conv_stem_pair_pool(d, (const float*)x, oi.in_c, oi.perm, oi.scale, oi.shift, tp.H, tp.W, st);
                    } else {
                        d.out = tptr(o.out);
                        rc = gdt_launch_conv_stem_pair(d, (const float*)x, oi.in_c, oi.perm, oi.scale, oi.shift, st);
                    }
                    if (net->profiling) net->last_variant[stp.op] = stp.pool_into >= 0 ? 952049 : 951000 + o.phases[0].ntaps;
                    break;
                }
                if (stp.xexp) {                // 3x3 + expand 1x1 + residual of a Bottleneck in one launch (conv3x3_expand_rb.hip)
                    const Op& oc = net->ops[stp.op + 1];
                    ConvLaunch d{};
                    conv_geometry(net, o, o.phases[0], n, ti, d);
                    d.in = tptr(o.in); d.zeros = zeros; d.relu = 1;
                    d.w_frag = (const f16*)(net->dev_blob + o.phases[0].w_frag_off);
                    d.bias = (const float*)(net->dev_blob + o.bias_off);
                    d.x_w_frag = (const f16*)(net->dev_blob + oc.phases[0].w_frag_off);
                    d.x_bias = (const f16*)(net->dev_blob + oc.bias_frag_off);          // (the bias as a weight fragment)
                    d.x_cout = oc.cd.cout;
                    d.res = tptr(oc.res); d.out = tptr(oc.out);
                    d.group_factor = c.group_factor;
                    GDT_REQUIRE(gdt_conv3x3_expand_eligible(d), "planned 3x3 + expand launch is not eligible at run time");
                    if (stp.xchain >= 0) {
                        const Op& a2 = net->ops[stp.xchain];
                        d.r_w_frag = (const f16*)(net->dev_blob + a2.phases[0].w_frag_off);
                        d.r_bias = (const float*)(net->dev_blob + a2.bias_off);
                        d.r_out = tptr(a2.out);
                    }
                    rc = gdt_launch_conv3x3_expand(d, st);
                    if (net->profiling && book) {
                        net->last_variant[stp.op] = (stp.xchain >= 0 ? 938000 : 939000) + oc.cd.cout / 8;
                        net->last_flops[stp.op] += net->last_flops[stp.op + 1]; net->last_flops[stp.op + 1] = 0.0;
                        // bytes: the 256-channel tensor between the two convs is neither written nor read
                        const double mid = (double)n * T[o.out].H * T[o.out].W * T[o.out].C * (double)net->esize();
                        net->last_bytes[stp.op] += net->last_bytes[stp.op + 1] - 2.0 * mid; net->last_bytes[stp.op + 1] = 0.0;
                        if (stp.xchain >= 0) {  // the chained reduce conv: its FLOPs, its output and weights -- its input is the tensor this launch has just written (not counted twice)
                            const Op& a2 = net->ops[stp.xchain];
                            const double yb = (double)n * T[a2.in].H * T[a2.in].W * T[a2.in].C * (double)net->esize();
                            net->last_flops[stp.op] += net->last_flops[stp.xchain]; net->last_flops[stp.xchain] = 0.0;
                            net->last_bytes[stp.op] += net->last_bytes[stp.xchain] - yb; net->last_bytes[stp.xchain] = 0.0;
                        }
                    }
                    break;
                }
                if (stp.kcat) {                // expand conv + its projection shortcut as one K-concatenated 1x1 GEMM (conv1x1_rb.hip)
                    const Op& ds = net->ops[o.kcat_ds];
                    ConvLaunch d{};
                    conv_geometry(net, o, o.phases[0], n, ti, d);
                    d.in = tptr(o.in); d.out = tptr(o.out); d.zeros = zeros;
                    d.w_frag = (const f16*)(net->dev_blob + o.kcat_frag_off);
                    d.bias = (const float*)(net->dev_blob + o.kcat_bias_off);
                    d.Kpad = o.cin_pad + ds.cin_pad; d.nk = d.Kpad / 64;
                    d.in2 = tptr(ds.in); d.in2_cin = ds.cin_pad; d.in2_h = T[ds.in].H; d.in2_w = T[ds.in].W; d.in2_stride = ds.cd.stride;
                    if (defer) { defer->kind = DEFER_CONV; defer->kcat = true; defer->d = d; }
                    else rc = gdt_launch_conv_1x1_rb(d, st);
                    if (net->profiling && book) {
                        net->last_variant[stp.op] = 946128;
                        net->last_flops[stp.op] += net->last_flops[o.kcat_ds]; net->last_flops[o.kcat_ds] = 0.0;
                        // bytes: the projected tensor is neither written (projection op) nor read back as the residual (expand op)
                        const double proj = (double)n * T[o.out].H * T[o.out].W * T[o.out].C * (double)net->esize();
                        net->last_bytes[stp.op] += net->last_bytes[o.kcat_ds] - 2.0 * proj; net->last_bytes[o.kcat_ds] = 0.0;
                    }
                    break;
                }
                ConvLaunch d{};
                d.in = tptr(o.in);
                if (stp.norm_from >= 0) {      // InstanceNorm(+ReLU) of the producer applied while staging the input
                    const Op& nj = net->ops[stp.norm_from];
                    d.in = tptr(nj.in);
                    d.in_norm = (const float*)(ws + plan.steps[stp.norm_from].aux_off[1]);
                    d.in_relu = nj.relu;
                    if (nj.res >= 0) d.in_res = tptr(nj.res);
                    if (plan.steps[stp.norm_from].wb) d.in_out = tptr(nj.out);
                    if (net->profiling) {       // bytes the folded form must move on top of the conv's own: the residual tensor read, the normalised tensor written back
                        const double tb = (double)n * T[nj.in].H * T[nj.in].W * T[nj.in].C * (double)net->esize();
                        net->last_bytes[stp.op] += tb * ((nj.res >= 0 ? 1.0 : 0.0) + (plan.steps[stp.norm_from].wb ? 1.0 : 0.0));
                    }
                }
                d.res = o.res >= 0 ? tptr(o.res) : nullptr;
                d.zeros = zeros;
                d.stats = stp.fused_stats ? (float*)(ws + stp.aux_off[0]) : nullptr;
                if (stp.ctf) {                  // transposed conv as one fused-phase GEMM (conv_igemm_rb.hip)
                    ctf_geometry(net, o, n, ti, d);
                    d.bias = o.has_bias ? (const float*)(net->dev_blob + o.ctf_bias_off) : nullptr;
                    d.out = tptr(o.out); d.out_f32 = nullptr;
                    d.w = nullptr; d.w_lo = nullptr; d.w_frag = o.ctf.has_frag ? (const f16*)(net->dev_blob + o.ctf.w_frag_off) : nullptr;
                    d.stats_tile_base = 0;
                    int variant = 960256;
                    if (net->precision == 2) {
                        d.w_cfrag = net->dev_blob + o.ctf.wc_off; d.wmx_a = net->dev_blob + o.ctf.wmx_a_off; d.wmx_b = net->dev_blob + o.ctf.wmx_b_off; d.wmx_s = net->dev_blob + o.ctf.wmx_s_off;
                        d.c_lo_exp = 12; d.c_hi_exp = 0;
                        variant = 980256;
                        rc = gdt_launch_conv_halo_c_ct(d, st);
                    } else
                    rc = gdt_conv_halo_ct_eligible(d) ? gdt_launch_conv_halo_ct(d, st) : gdt_launch_conv_igemm_rb(d, st, &variant);
                    if (net->profiling) net->last_variant[stp.op] = variant;
                    break;
                }
                if (stp.s2) {                   // stride-2 conv as the shift form over the virtual space-to-depth input (conv3x3_halo_c.hip)
                    s2_geometry(net, o, n, ti, d);
                    d.bias = o.has_bias ? (const float*)(net->dev_blob + o.s2_bias_off) : nullptr;
                    d.out = tptr(o.out); d.out_f32 = nullptr; d.w = nullptr; d.w_lo = nullptr; d.w_frag = nullptr;
                    d.stats_tile_base = 0;
                    if (net->precision == 1) {      // f16x3: the patch kernel over the same view (conv3x3_halo_x3.hip FORM 2)
                        d.w = (const f16*)(net->dev_blob + o.s2.w_off); d.w_lo = (const f16*)(net->dev_blob + o.s2.w_lo_off); d.x3_form = 2;
                        int variant = 0;
                        rc = gdt_launch_conv_x3(d, st, &variant);
                        if (net->profiling) net->last_variant[stp.op] = variant;
                        break;
                    }
                    d.w_cfrag = net->dev_blob + o.s2.wc_off; d.wmx_a = net->dev_blob + o.s2.wmx_a_off; d.wmx_b = net->dev_blob + o.s2.wmx_b_off; d.wmx_s = net->dev_blob + o.s2.wmx_s_off;
                    d.c_lo_exp = 12; d.c_hi_exp = 0;
                    d.stats_tile_base = 0;
                    rc = gdt_launch_conv_halo_c_s2(d, st);
                    if (net->profiling) net->last_variant[stp.op] = 990256;
                    break;
                }
                if (stp.ctp) {                  // f16x3 transposed conv, 64 output channels: the phases (py, 0) and (py, 1) as one 128-column launch per py
                    int pi = 0;
                    for (const PackedPhase& pp : o.pairs) {
                        pair_geometry(net, o, pp, n, ti, d);
                        d.bias = o.has_bias ? (const float*)(net->dev_blob + o.pair_bias_off) : nullptr;
                        d.out = tptr(o.out); d.out_f32 = nullptr;
                        d.w = (const f16*)(net->dev_blob + pp.w_off); d.w_lo = (const f16*)(net->dev_blob + pp.w_lo_off); d.w_frag = nullptr;
                        d.stats_tile_base = 2 * pi * (d.M / 128);          // (record sets in phase order (0,0) (0,1) (1,0) (1,1): the kernel puts the second half one set further on)
                        ++pi;
                        int variant = 0;
                        rc = gdt_launch_conv_x3(d, st, &variant);
                        if (net->profiling) net->last_variant[stp.op] = variant;
                        if (rc != GDT_OK) break;
                    }
                    break;
                }
                int phase_idx = 0;
                bool fused_head = false;
                for (const PackedPhase& ph : o.phases) {
                    conv_geometry(net, o, ph, n, ti, d);
                    d.bias = o.has_bias ? (const float*)(net->dev_blob + o.bias_off) : nullptr;
                    if (o.rowsplit) { d.out = (f16*)(ws + stp.aux_off[1]); d.out_f32 = nullptr; d.Cout = o.rs_cout8; d.bias = nullptr; }
                    else if (o.cd.out_f32_nchw) { d.out = nullptr; d.out_f32 = (float*)outputs[o.slot]; }
                    else if (stp.pool_into >= 0) { d.out = tptr(net->ops[stp.pool_into].out); d.out_f32 = nullptr; d.pool2 = 1; }
                    else { d.out = tptr(o.out); d.out_f32 = nullptr; }
                    d.w = (const f16*)(net->dev_blob + ph.w_off);
                    d.w_lo = f32 ? (const f16*)(net->dev_blob + ph.w_lo_off) : nullptr;
                    d.w_frag = ph.has_frag ? (const f16*)(net->dev_blob + ph.w_frag_off) : nullptr;
                    if (ph.has_mx) {
                        d.w_cfrag = net->dev_blob + ph.wc_off; d.wmx_a = net->dev_blob + ph.wmx_a_off; d.wmx_b = net->dev_blob + ph.wmx_b_off; d.wmx_s = net->dev_blob + ph.wmx_s_off;
                        d.c_lo_exp = 12; d.c_hi_exp = 0;
                    }
                    if (ph.has_mx16) {
                        d.w_c16 = net->dev_blob + ph.w16_off;
                    }
                    d.stats_tile_base = phase_idx * (d.M / 128);
                    ++phase_idx;
                    int variant = 0;
                    if (o.rowsplit) {           // fused head kernel (f16c: fp32 input, rounded once while staging; f16x3: split twice): GEMM over the kernel rows + combine + activation in one launch
                        ConvLaunch h = d;
                        h.out = nullptr; h.out_f32 = (float*)outputs[o.slot]; h.Cout = o.cd.cout; h.act = o.cd.act; h.in_f32 = f32;
                        h.w_frag2 = (net->precision == 1 && ph.has_frag) ? (const f16*)(net->dev_blob + ph.w_frag2_off) : nullptr;
                        h.bias = o.has_bias ? (const float*)(net->dev_blob + o.rs_bias_off) : nullptr;
                        if (gdt_conv_head7_eligible(h)) {
                            rc = gdt_launch_conv_head7(h, st);
                            if (net->profiling) net->last_variant[stp.op] = 920007;
                            fused_head = true;
                            break;
                        }
                    }
                    if (stp.aug) { d.w_frag2 = (const f16*)(net->dev_blob + ph.w_frag2_off); variant = 955000 + ph.ntaps; rc = gdt_launch_conv_stem_c(d, st); }
                    else if (net->precision == 2 && gdt_conv_halo_c16_eligible(d)) { variant = 971256; rc = gdt_launch_conv_halo_c16(d, st); }
                    else if (net->precision == 2 && gdt_conv_halo_c_eligible(d)) { variant = 970000 + gdt_conv_halo_c_columns(d); rc = gdt_launch_conv_halo_c(d, st); }
                    else if (defer && !f32 && o.phases.size() == 1 && !o.rowsplit && !o.cd.out_f32_nchw) { defer->kind = DEFER_CONV; defer->kcat = false; defer->d = d; }
                    else rc = f32 ? gdt_launch_conv_x3(d, st, &variant) : gdt_launch_conv(d, st, &variant);
                    if (net->profiling) net->last_variant[stp.op] = variant;
                    if (rc != GDT_OK) break;
                }
                if (o.rowsplit && !fused_head && rc == GDT_OK)
                    rc = gdt_k_rowsplit_combine(d.out, f32, o.has_bias ? (const float*)(net->dev_blob + o.rs_bias_off) : nullptr,
                                                (float*)outputs[o.slot], n, d.OH, ti.W, o.rs_cout8, o.cd.cout, o.cd.kw, o.cd.pad,
                                                o.cd.pad_reflect, o.cd.act, st);
                break;
            }
            case OP_INORM: {
                const Tensor& ti = T[o.in];
                if (stp.norm_into >= 0)     // the consuming conv applies it: only mean / rstd are produced here
                    rc = gdt_k_instance_norm_stats(tptr(o.in), f32, stp.fused_stats, (float*)(ws + stp.aux_off[0]), stp.tiles_per_image,
                                                   stp.fused_stats ? plan.steps[o.stats_from].stats_sets : 1,
                                                   (float*)(ws + stp.aux_off[1]), n, ti.H * ti.W, ti.C, o.eps, st);
                else if (stp.fused_stats)
                    rc = gdt_k_instance_norm_fused(tptr(o.in), o.res >= 0 ? tptr(o.res) : nullptr, tptr(o.out), f32,
                                                   (const float*)(ws + stp.aux_off[0]), stp.tiles_per_image,
                                                   plan.steps[o.stats_from].stats_sets, (float*)(ws + stp.aux_off[1]), n,
                                                   ti.H * ti.W, ti.C, o.eps, o.relu, st);
                else
                    rc = gdt_k_instance_norm(tptr(o.in), o.res >= 0 ? tptr(o.res) : nullptr, tptr(o.out), f32, (float*)(ws + stp.aux_off[0]),
                                             (float*)(ws + stp.aux_off[1]), n, ti.H * ti.W, ti.C, o.eps, o.relu, st);
                break;
            }
            case OP_MAXPOOL: {
                if (stp.skip) break;                  // done by the producing conv's epilogue
                const Tensor& ti = T[o.in]; const Tensor& to = T[o.out];
                rc = gdt_k_maxpool(tptr(o.in), tptr(o.out), f32, n, ti.H, ti.W, ti.C, to.H, to.W, o.k, o.s, o.p, st);
                break;
            }
            case OP_GEM: {
                const Tensor& ti = T[o.in];
                rc = gdt_k_gem_l2n(tptr(o.in), f32, (float*)(ws + stp.aux_off[0]), (float*)outputs[o.slot], n, ti.H * ti.W, ti.C, o.gem_p,
                                   o.eps_gem, o.eps_l2, st);
                break;
            }
            case OP_OUT_NCHW: {
                const Tensor& ti = T[o.in];
                rc = gdt_k_unpack_output(tptr(o.in), f32, (float*)outputs[o.slot],
                                         o.tap_has_bias ? (const float*)(net->dev_blob + o.tap_bias_off) : nullptr, n, ti.H * ti.W, ti.C, st);
                break;
            }
            case OP_HED: {
                const float* sc[5]; int hh[5], wwv[5];
                for (int k = 0; k < 5 && rc == GDT_OK; ++k) {
                    const Tensor& tf = T[o.feats[k]];
                    float* s = (float*)(ws + stp.aux_off[k]);
                    rc = gdt_k_hed_score(tptr(o.feats[k]), f32, (const float*)(net->dev_blob + o.score_w_off[k]), o.score_b[k], s,
                                         (long)n * tf.H * tf.W, tf.C, st);
                    sc[k] = s; hh[k] = tf.H; wwv[k] = tf.W;
                }
                if (rc == GDT_OK) rc = gdt_k_hed_fuse(sc, hh, wwv, o.fusion_w, o.fusion_b, (float*)outputs[o.slot], n, rh, rw, o.sigmoid, st);
                break;
            }
        }
    return rc;
}

// weights of a fused Bottleneck step (the same for every geometry)
int launch_bneck_levels(gdt_net* net, const Step& stp, const Deferred* df, int L, hipStream_t st) {
    const bool dsf = stp.bneck_ds >= 0;
    const Op &oa = net->ops[stp.bneck_a], &ob = net->ops[stp.op + (dsf ? 2 : 1)], &oc = net->ops[stp.op + (dsf ? 3 : 2)];
    const Op* od = dsf ? &net->ops[stp.bneck_ds] : nullptr;
    const f16* xs[GDT_MAX_LEVELS]; f16* ys[GDT_MAX_LEVELS]; int ns[GDT_MAX_LEVELS], hs[GDT_MAX_LEVELS], wsz[GDT_MAX_LEVELS];
    for (int l = 0; l < L; ++l) { xs[l] = df[l].bx; ys[l] = df[l].by; ns[l] = df[l].bn; hs[l] = df[l].bh; wsz[l] = df[l].bw; }
    return gdt_launch_bneck_levels(xs, ys, (const f16*)(net->dev_blob + oa.phases[0].w_frag_off), (const f16*)(net->dev_blob + ob.phases[0].w_frag_off),
                                   (const f16*)(net->dev_blob + oc.phases[0].w_frag_off), (const float*)(net->dev_blob + oa.bias_off),
                                   (const float*)(net->dev_blob + ob.bias_off), (const float*)(net->dev_blob + oc.bias_off),
                                   od ? (const f16*)(net->dev_blob + od->phases[0].w_frag_off) : nullptr, od ? (const float*)(net->dev_blob + od->bias_off) : nullptr,
                                   oa.cd.cin, oc.cd.cout, oa.cd.cout, ns, hs, wsz, L, st);
}

// The forward on L independent geometries in lock-step: op by op, every geometry's launch of the op -- joined into ONE launch where the kernel has a
// multi-geometry entry and the levels select the same kernel family (conv1x1_rb.hip, conv3x3_halo_rb.hip, conv_bneck.hip), else issued one after the other
int forward_levels(gdt_net* net, LevelCtx* cx, int L, hipStream_t st) {
    const int nops = (int)net->ops.size();
    net->last_joined = 0; net->last_level_launches = 0;
    if (net->profiling) {
        net->last_flops.assign(nops, 0.0);
        net->last_variant.assign(nops, 0);
        net->last_bytes.assign(nops, 0.0);
        for (int l = 0; l < L; ++l) {
            net->tensors = cx[l].T;                    // (op_flops / op_bytes read the planned shapes from the net's table)
            for (int i = 0; i < nops; ++i) {
                net->last_flops[i] += op_flops(net, net->ops[i], cx[l].n, cx[l].rh, cx[l].rw);
                net->last_bytes[i] += op_bytes(net, net->ops[i], cx[l].n);
            }
        }
    }
    for (int i = 0; i < nops; ++i) {
        if (net->profiling) GDT_CHECK_HIP(hipEventRecord(net->events[2 * i], st));
        int rc = GDT_OK;
        if (L == 1) rc = exec_step(net, cx[0], cx[0].plan.steps[i], st, nullptr, true);
        else {
            Deferred df[GDT_MAX_LEVELS];
            int nconv = 0, nbneck = 0, nkcat = 0;
            for (int l = 0; l < L && rc == GDT_OK; ++l) {
                rc = exec_step(net, cx[l], cx[l].plan.steps[i], st, &df[l], l == 0);
                nconv += df[l].kind == DEFER_CONV; nbneck += df[l].kind == DEFER_BNECK; nkcat += df[l].kind == DEFER_CONV && df[l].kcat;
            }
            if (rc != GDT_OK) return rc;
            net->last_level_launches += nconv + nbneck;
            bool joined = false;
            if (nbneck == L) { rc = launch_bneck_levels(net, cx[0].plan.steps[i], df, L, st); joined = true; }
            else if (nconv == L && (nkcat == 0 || nkcat == L)) {
                ConvLaunch dl[GDT_MAX_LEVELS];
                int fam = nkcat ? 1 : gdt_conv_family(df[0].d);
                for (int l = 0; l < L; ++l) { dl[l] = df[l].d; if (!nkcat && gdt_conv_family(df[l].d) != fam) fam = 0; }
                if (fam == 1) { rc = gdt_launch_conv_1x1_rb_levels(dl, L, st); joined = true; if (net->profiling && !nkcat) net->last_variant[i] = 945128; }
                else if (fam == 2 && gdt_conv_halo_rb_levels_ok(dl, L)) { rc = gdt_launch_conv_halo_rb_levels(dl, L, st); joined = true; if (net->profiling) net->last_variant[i] = 910256; }
            }
            if (joined) ++net->last_joined;
            static const bool lv_dbg = getenv("GDT_LEVELS_DEBUG") != nullptr;
            if (lv_dbg && (nconv || nbneck)) {
                fprintf(stderr, "[levels] op %d joined %d:", i, (int)joined);
                for (int l = 0; l < L; ++l) {
                    if (df[l].kind == DEFER_CONV) fprintf(stderr, " [conv%s fam %d M %d Cin %d Cout %d taps %d s%d]", df[l].kcat ? " kcat" : "", gdt_conv_family(df[l].d), df[l].d.M, df[l].d.Cin, df[l].d.Cout, df[l].d.ntaps, df[l].d.sy);
                    else if (df[l].kind == DEFER_BNECK) fprintf(stderr, " [bneck %dx%dx%d]", df[l].bn, df[l].bh, df[l].bw);
                    else fprintf(stderr, " [-]");
                }
                fprintf(stderr, "\n");
            }
            if (!joined) {                    // one by one (levels whose step was not handed back have launched already)
                for (int l = 0; l < L && rc == GDT_OK; ++l) {
                    if (df[l].kind == DEFER_BNECK) rc = launch_bneck_levels(net, cx[l].plan.steps[i], &df[l], 1, st);
                    else if (df[l].kind == DEFER_CONV && df[l].kcat) rc = gdt_launch_conv_1x1_rb(df[l].d, st);
                    else if (df[l].kind == DEFER_CONV) { int variant = 0; rc = gdt_launch_conv(df[l].d, st, &variant); if (net->profiling && l == 0) net->last_variant[i] = variant; }
                }
            }
        }
        if (rc != GDT_OK) return rc;
        if (net->profiling) GDT_CHECK_HIP(hipEventRecord(net->events[2 * i + 1], st));
    }
    return GDT_OK;
}

int plan_level(gdt_net* net, LevelCtx& c, void* workspace, size_t workspace_bytes) {
    int rc = make_plan(net, c.n, c.rh, c.rw, c.plan, c.rh == c.h && c.rw == c.w);
    if (rc != GDT_OK) return rc;
    if (c.plan.peak + ALIGN > workspace_bytes || !workspace) {
        gdt_set_error("workspace too small: need " + std::to_string(c.plan.peak + ALIGN) + " bytes, got " + std::to_string(workspace_bytes));
        return GDT_ERR_WORKSPACE;
    }
    c.ws = (char*)(((uintptr_t)workspace + ALIGN - 1) / ALIGN * ALIGN);
    c.T = net->tensors;
    c.group_factor = net->group_factor;
    return GDT_OK;
}

}  // namespace

extern "C" {

int gdt_net_forward(gdt_net* net, const float* x, int n, int h, int w, int rh, int rw, float rscale,
                    void* const* outputs, int n_outputs, void* workspace, size_t workspace_bytes, void* stream) {
    GDT_REQUIRE(net && net->finalized, "net must be finalized");
    GDT_REQUIRE(x && n >= 1 && h >= 1 && w >= 1 && rh >= 1 && rw >= 1, "input geometry");
    GDT_REQUIRE(n_outputs == (int)net->out_ops.size() && (outputs || n_outputs == 0), "output count");
    GDT_REQUIRE((long)n * rh * rw < (1l << 31) && (long)n * h * w < (1l << 31), "N*H*W must stay below 2^31");
    for (int i = 0; i < n_outputs; ++i) GDT_REQUIRE(outputs[i] != nullptr, "null output buffer");
    LevelCtx c;
    c.x = x; c.n = n; c.h = h; c.w = w; c.rh = rh; c.rw = rw; c.rscale = rscale; c.outputs = outputs;
    int rc = plan_level(net, c, workspace, workspace_bytes);
    if (rc != GDT_OK) return rc;
    return forward_levels(net, &c, 1, (hipStream_t)stream);
}

int gdt_net_set_group_factor(gdt_net* net, float factor) {
    GDT_REQUIRE(net && factor >= 1.f && factor < 1e6f, "group factor >= 1");
    net->group_factor = factor;
    return GDT_OK;
}

int gdt_net_levels_joined(gdt_net* net, int* level_launches) {
    if (!net) return 0;
    if (level_launches) *level_launches = net->last_level_launches;
    return net->last_joined;
}

int gdt_net_forward_levels(gdt_net* net, const gdt_level* levels, int n_levels, void* stream) {
    GDT_REQUIRE(net && net->finalized, "net must be finalized");
    GDT_REQUIRE(levels && n_levels >= 1 && n_levels <= GDT_MAX_LEVELS, "1..4 geometries per call");
    std::vector<LevelCtx> cx(n_levels);
    double group_px = 0.0;
    for (int l = 0; l < n_levels; ++l) group_px += (double)levels[l].n * levels[l].rh * levels[l].rw;
    const float saved_factor = net->group_factor;
    struct Restore { gdt_net* n; float f; ~Restore() { n->group_factor = f; } } restore{net, saved_factor};
    for (int l = 0; l < n_levels; ++l) {
        const gdt_level& g = levels[l];
        if (n_levels > 1 && g.n >= 1 && g.rh >= 1 && g.rw >= 1) net->group_factor = (float)(group_px / ((double)g.n * g.rh * g.rw));
        GDT_REQUIRE(g.x && g.n >= 1 && g.h >= 1 && g.w >= 1 && g.rh >= 1 && g.rw >= 1, "input geometry");
        GDT_REQUIRE(g.n_outputs == (int)net->out_ops.size() && (g.outputs || g.n_outputs == 0), "output count");
        GDT_REQUIRE((long)g.n * g.rh * g.rw < (1l << 31) && (long)g.n * g.h * g.w < (1l << 31), "N*H*W must stay below 2^31");
        for (int i = 0; i < g.n_outputs; ++i) GDT_REQUIRE(g.outputs[i] != nullptr, "null output buffer");
        for (int k = 0; k < l; ++k) {      // every geometry its own scratch memory
            const char *a0 = (const char*)levels[k].workspace, *a1 = a0 + levels[k].workspace_bytes, *b0 = (const char*)g.workspace, *b1 = b0 + g.workspace_bytes;
            GDT_REQUIRE(a1 <= b0 || b1 <= a0, "the geometries of one call need disjoint workspaces");
        }
        LevelCtx& c = cx[l];
        c.x = g.x; c.n = g.n; c.h = g.h; c.w = g.w; c.rh = g.rh; c.rw = g.rw; c.rscale = g.rscale; c.outputs = g.outputs;
        const int rc = plan_level(net, c, g.workspace, g.workspace_bytes);
        if (rc != GDT_OK) return rc;
    }
    return forward_levels(net, cx.data(), n_levels, (hipStream_t)stream);
}

int gdt_ms_aggregate(const float* x, float* y, int scales, int n, int d, float msp, void* stream) {
    GDT_REQUIRE(x && y && scales >= 1 && n >= 1 && d >= 1, "ms_aggregate arguments");
    return gdt_k_ms_aggregate(x, y, scales, n, d, msp, (hipStream_t)stream);
}

int gdt_whiten(const float* P, const float* m, const float* v, float* tmp, float* out, int n, int d, int dims, void* stream) {
    GDT_REQUIRE(P && m && v && tmp && out && n >= 1 && d >= 1 && dims >= 1 && dims <= d, "whiten arguments");
    return gdt_k_whiten(P, m, v, tmp, out, n, d, dims, (hipStream_t)stream);
}

int gdt_whiten_f64(const double* P, const double* m, const double* v, double* tmp, double* out, int n, int d, int dims, void* stream) {
    GDT_REQUIRE(P && m && v && tmp && out && n >= 1 && d >= 1 && dims >= 1 && dims <= d, "whiten arguments");
    return gdt_k_whiten_f64(P, m, v, tmp, out, n, d, dims, (hipStream_t)stream);
}

int gdt_gem_l2n(const float* fmap, int n, int d, int h, int w, float p, float eps_gem, float eps_l2, float* pooled, float* out, void* stream) {
    GDT_REQUIRE(fmap && pooled && out && n >= 1 && d >= 1 && h >= 1 && w >= 1 && p > 0.f, "gem_l2n arguments");
    GDT_REQUIRE((long)h * w < (1l << 31), "feature map too large");
    return gdt_k_gem_l2n_nchw(fmap, pooled, out, n, d, h * w, p, eps_gem, eps_l2, (hipStream_t)stream);
}

int gdt_l2n_rows(const float* x, float* y, int n, int d, float eps, void* stream) {
    GDT_REQUIRE(x && y && n >= 1 && d >= 1, "l2n arguments");
    return gdt_k_l2n_rows(x, y, n, d, eps, (hipStream_t)stream);
}

}  // extern "C"
