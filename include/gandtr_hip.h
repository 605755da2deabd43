/* gandtr_hip.h -- C ABI of libgandtr_hip.so, the MI355X (gfx950) implementation of the gandtr inference hot path.
 *
 * The reference (mohwald/gandtr) is pure Python on torch and has NO native boundary of its own; every entry point
 * below therefore replaces a *Python-level* call site of the reference, cited as file:line relative to the reference
 * root.  The host side (gandtr_amd/, Python) binds these symbols with ctypes (gandtr_amd/_hip.py) -- see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are raw HIP device addresses (e.g. torch.Tensor.data_ptr()).
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); NULL = default stream.
 *   - every function returns 0 (GDT_OK) or a positive status; gdt_last_error() returns the thread-local message.
 *     The Python binding re-raises GDT_ERR_INVALID as ValueError / AssertionError like the reference's own checks
 *     (SURVEY.md section 8b "Error conventions").
 *   - a gdt_net is bound to the device that was current at gdt_net_finalize(); host calls on one handle must be serialised (see gdt_net_forward).
 *   - ownership: the caller (PyTorch) owns all input / output / workspace buffers; a net owns its packed weights.
 *   - external images are fp32 NCHW (the reference's layout); internal activations are fp16 NHWC.
 */
#ifndef GANDTR_HIP_H
#define GANDTR_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDT_OK 0
#define GDT_ERR_INVALID 1   /* bad argument / unsupported shape   */
#define GDT_ERR_HIP 2       /* a HIP runtime call failed          */
#define GDT_ERR_WORKSPACE 3 /* workspace too small                */
#define GDT_ERR_NOT_CONVERGED 4 /* an iterative solver hit its cap */

const char* gdt_last_error(void);
/* library self-description: "gandtr_hip <version> gfx950" */
const char* gdt_version(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Network graph builder + executor.
 * Replaces nn.Sequential / nn.Module.forward execution of
 *   ResnetGenerator.forward          mdir/components/model/network/p2p_networks.py:315-337 (layers :269-313, :454-506)
 *   ImageRetrievalNet.forward        mdir/external/cirtorch/networks/imageretrievalnet.py:101-123
 *   HedInterpolation.forward         mdir/components/model/network/hed.py:60-83
 * Tensors are referred to by the integer ids the builder returns.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct gdt_net gdt_net;

int gdt_net_create(gdt_net** net);
void gdt_net_destroy(gdt_net* net);

/* Arithmetic of the conv GEMMs, to be chosen before the first op is added.
 *   0 "f16"   (default): fp16 NHWC activations, one fp16 MFMA pass, fp32 accumulation (11-bit operands).
 *   1 "f16x3": fp32 NHWC activations; every operand is split into two fp16 numbers and three MFMA passes are accumulated
 *              (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo): fp32-class accuracy at 3x the matrix work.  Used where the reference's
 *              fp32 result has to be matched to 1e-3 through deep random-weight stacks.
 *   2 "f16c":  fp32 NHWC activations; fp16 MFMA product + one block-scaled fp4 x fp6 MFMA per 32 k-values that carries both rounding
 *              residuals (1.5x the matrix work): 1e-3-class accuracy; the generators' default.  The last layer (7x7 head) runs a single
 *              fp16 pass on the fp32 tensor.
 *   3 "f16ch": f16c with the head compensated as well (second MFMA pass over an fp4 plane): pre-tanh error 4.7e-4 -> 3.1e-4 of the
 *              range for +0.5 ms per 64 x 256^2 batch. */
int gdt_net_set_precision(gdt_net* net, int mode);

/* External fp32 NCHW image input with C <= 8 channels (packed to fp16 NHWC8 on entry).  Optional channel permutation
 * (RgbToBgrPre, mdir/components/data/wrapper.py:351-364) and per-channel affine y = x*scale + shift
 * (MeanStdPost/Pre._adapt, wrapper.py:172-175); pass NULL for identity.  A bilinear resize
 * (F.interpolate(scale_factor=s, mode='bilinear', align_corners=False), wrapper.py:225) is selected per forward call. */
int gdt_net_input(gdt_net* net, int channels, const int* perm, const float* scale, const float* shift, int* out_tensor);

typedef struct gdt_conv_desc {
    int cin, cout;        /* logical channel counts (cin is padded to a power of two >= 8 internally)            */
    int kh, kw;           /* kernel size (1, 3 or 7 on the hot path)                                               */
    int stride;           /* 1 or 2                                                                                */
    int pad;              /* padding on every side                                                                 */
    int pad_reflect;      /* 0: zeros (Conv2d padding=p), 1: nn.ReflectionPad2d(p) in front of the conv            */
    int transposed;       /* 1: nn.ConvTranspose2d(k3, s2, p1, output_padding 1)  (p2p_networks.py:295-298)        */
    int relu;             /* fuse nn.ReLU after bias (+BN) (+residual)                                             */
    int out_f32_nchw;     /* 1: result is an EXTERNAL fp32 NCHW output (generator head), 0: internal fp16 tensor   */
    int act;              /* external output only: 0 none, 1 tanh (p2p_networks.py:311), 2 sigmoid                 */
    float bn_eps;         /* eps of the folded BatchNorm2d (1e-5)                                                  */
} gdt_conv_desc;

/* Conv2d / ConvTranspose2d with host fp32 weights in torch layout ([cout][cin][kh][kw], transposed: [cin][cout][kh][kw]).
 * bias and the four BatchNorm2d(eval) vectors may be NULL; BN is folded into the packed weights (get_norm_layer "batch",
 * p2p_networks.py:27; torchvision ResNet BN).  residual_tensor >= 0 adds that tensor before the ReLU
 * (ResnetBlock.forward p2p_networks.py:505; torchvision Bottleneck).  out_tensor receives the new tensor id, or the
 * external output slot index when out_f32_nchw = 1. */
int gdt_net_conv(gdt_net* net, int in_tensor, const gdt_conv_desc* desc, const float* weight, const float* bias,
                 const float* bn_gamma, const float* bn_beta, const float* bn_mean, const float* bn_var,
                 int residual_tensor, int* out_tensor);

/* nn.InstanceNorm2d(affine=False, eps) (+ fused ReLU) (+ fused residual add AFTER the norm): p2p_networks.py:29,:272,:505 */
int gdt_net_instance_norm(gdt_net* net, int in_tensor, float eps, int relu, int residual_tensor, int* out_tensor);

/* nn.MaxPool2d(kernel, stride, padding), floor mode */
int gdt_net_maxpool(gdt_net* net, int in_tensor, int kernel, int stride, int pad, int* out_tensor);

/* l2n(gem(x, p, eps_gem), eps_l2): cirtorch layers/functional.py:21-22,:130-131.  External output: fp32 [N][D]
 * row-major, i.e. the memory the reference's `o.permute(1,0)` view aliases (imageretrievalnet.py:123). */
int gdt_net_gem_l2n(gdt_net* net, int in_tensor, float p, float eps_gem, float eps_l2, int* out_slot);

/* Feature tap: internal fp16 NHWC tensor -> external fp32 NCHW (+ optional per-channel bias), p2p_networks.py:316-334 */
int gdt_net_output_nchw(gdt_net* net, int in_tensor, const float* bias, int* out_slot);

/* HED head (hed.py:67-83): five 1x1 score convs (weights score_w[k] of length channels of tensor k, bias score_b[k]),
 * bilinear upsampling of each score map to the network input size, 1x1 fusion (fusion_w[5], fusion_b), sigmoid.
 * External output fp32 [N][1][H][W]. */
int gdt_net_hed_head(gdt_net* net, const int* feature_tensors, const float* const* score_w, const float* score_b,
                     const float* fusion_w, float fusion_b, int sigmoid, int* out_slot);

/* Upload packed weights to the current device.  No more ops can be added afterwards. */
int gdt_net_finalize(gdt_net* net);

/* Shape of an external output for a given input geometry.  (rh, rw) is the resized input size (== h, w without
 * resize).  dims receives up to 4 ints, ndim their count. */
int gdt_net_output_shape(gdt_net* net, int slot, int n, int rh, int rw, int* dims, int* ndim);
int gdt_net_num_outputs(gdt_net* net);

/* Bytes of caller-provided scratch needed by gdt_net_forward for this geometry. */
int gdt_net_workspace_bytes(gdt_net* net, int n, int rh, int rw, size_t* bytes);

/* Run the graph.  x: device fp32 [n][c][h][w].  If (rh, rw) != (h, w) the input is bilinearly resized with torch's
 * scale_factor semantics, rscale = (float)(1.0 / scale_factor).  outputs[i] is the device buffer of external slot i. */
/* Threading: host calls on one handle must be serialised (each call plans its geometry into the handle before it enqueues its launches); all
 * per-forward DEVICE state lives in the caller's workspace, so forwards of one handle may overlap on the device when each is given its own
 * workspace and stream (HipNet.forward_many does that for the levels of the multi-scale pyramid). */
int gdt_net_forward(gdt_net* net, const float* x, int n, int h, int w, int rh, int rw, float rscale,
                    void* const* outputs, int n_outputs, void* workspace, size_t workspace_bytes, void* stream);

/* The forward on several independent geometries at once -- the levels of a multi-scale pyramid (CirMultiscaleAggregation.preprocess, mdir/components/data/wrapper.py:221-233,
 * whose levels the reference runs one after the other, mdir/learning/network.py:139-140), or equal-shaped groups of a list of images.  Every level has its own input,
 * (resized) size, outputs and DISJOINT workspace (sized by gdt_net_workspace_bytes for its geometry).  The ops run in lock-step; where the kernel has a
 * multi-geometry entry (1x1 convs, 3x3 patch convs, fused Bottlenecks) the levels' launches of an op are ONE launch, whose workgroups are dealt out to the levels.
 * Results are those of gdt_net_forward called level by level, bit for bit. */
typedef struct gdt_level {
    const float* x; int n, h, w, rh, rw; float rscale;      /* as the arguments of gdt_net_forward */
    void* const* outputs; int n_outputs;
    void* workspace; size_t workspace_bytes;
} gdt_level;
int gdt_net_forward_levels(gdt_net* net, const gdt_level* levels, int n_levels, void* stream);
/* Planner hint for geometries that run CONCURRENTLY on one device (the levels of a pyramid issued on side streams): factor = (pixels of all geometries in flight) /
 * (pixels of the geometry planned next), >= 1; 1 = alone (the default).  Fusions whose tile thresholds mean "enough patches to fill the chip" then count the
 * group's patches.  It applies to every later plan (gdt_net_workspace_bytes, gdt_net_output_shape, gdt_net_plan_summary, gdt_net_forward) until set again;
 * gdt_net_forward_levels sets it per level itself.  Host calls on one handle are serialised (see the threading note above). */
int gdt_net_set_group_factor(gdt_net* net, float factor);
/* diagnostics of the last gdt_net_forward_levels call: returns the number of ops whose levels ran as ONE launch; *level_launches = launches the levels handed
 * to the lock-step driver in total (joined or not) */
int gdt_net_levels_joined(gdt_net* net, int* level_launches);

/* Algorithmic conv FLOPs (2*MACs, real channel counts, bias/norm/activation excluded) of one forward at this geometry:
 * the numerator of bench.py's roofline.achieved. */
int gdt_net_flops(gdt_net* net, int n, int rh, int rw, double* flops);

/* The planner's decisions for a geometry as counts (host logic only, no device call): counts[0] conv launches (a fused launch counts once), [1] whole Bottlenecks in one
 * launch, [2] 3x3 + expand launches, [3] of those with the next block's reduce conv chained in, [4] projection shortcuts folded into their expand conv, [5] InstanceNorms
 * applied by their consumer's staging, [6] max-pools written by their producer, [7] 1 if the stem reads the caller's image itself (resize = 0 only), [8] transposed convs as
 * one fused-phase launch, [9] stride-2 convs as the shift form.  n_counts >= 10.  (Diagnostics / tests; no reference counterpart.) */
int gdt_net_plan_summary(gdt_net* net, int n, int rh, int rw, int resize, int* counts, int n_counts);

/* Per-op timing for bench.py's roofline line: when enabled, gdt_net_forward records HIP events on the caller's stream
 * around every op.  gdt_net_profile_read (after the forward) returns per op: kind (0 input, 1 conv, 2 instance-norm,
 * 3 maxpool, 4 gem, 5 tap, 6 hed), the conv kernel variant (BM*1000+BN: conv_igemm_kernel<BM,BN>; 900000+BN:
 * conv3x3_halo_kernel<BN>), elapsed ms and the algorithmic FLOPs.  No reference counterpart (the reference has wall-clock StopWatch only, mdir/tools/stats.py:48-68). */
int gdt_net_set_profiling(gdt_net* net, int enable);
int gdt_net_profile_read(gdt_net* net, int max_ops, int* n_ops, int* kinds, int* tile_n, double* ms, double* flops);
/* ... and the ALGORITHMIC HBM bytes of each op of that forward (conv ops: input once -- a strided 1x1 conv only the pixels it samples --,
 * output once, residual once, fp16 weights once; a fused launch is booked on its first op without the tensors that never exist; a conv that
 * applies its producer's InstanceNorm while staging also counts the block residual it adds and the normalised tensor it writes back):
 * the numerator of bench.py's HBM view of the Bottleneck 1x1 convs (SURVEY.md section 8d "compulsory bytes"). */
int gdt_net_profile_read_bytes(gdt_net* net, int max_ops, int* n_ops, double* bytes);
/* number of ops of the graph (= entries the two profile readers return; size the buffers with it) */
int gdt_net_num_ops(gdt_net* net);

/* ------------------------------------------------------------------------------------------------------------------
 * Stand-alone descriptor ops (device fp32 buffers)
 * ------------------------------------------------------------------------------------------------------------------ */

/* CirMultiscaleAggregation.aggregate_tensor (wrapper.py:236-245), batched over images: x [S][N][D] -> y [N][D],
 * y = (mean_s x^msp)^(1/msp), then y /= ||y||_2 (no eps). */
int gdt_ms_aggregate(const float* x, float* y, int scales, int n, int d, float msp, void* stream);

/* CirtorchWhiten.postprocess (wrapper.py:320-322), batched: v [N][D], P [D][D] row-major, m [D] -> out [N][dims],
 * out = P[:dims] (v - m) / (||.||_2 + 1e-6).  tmp: [N][dims] scratch. */
int gdt_whiten(const float* P, const float* m, const float* v, float* tmp, float* out, int n, int d, int dims, void* stream);
/* the same in float64: the arithmetic of the reference's `whiten` STAGE (mdir/stages/whiten.py:20-23 -> whitenapply,
 * mdir/external/cirtorch/utils/whiten.py:4-12: numpy promotes against the float64 P) */
int gdt_whiten_f64(const double* P, const double* m, const double* v, double* tmp, double* out, int n, int d, int dims, void* stream);

/* LF.gem + LF.l2n on an fp32 NCHW feature map (cirtorch layers/functional.py:21-22, :130-131; the tail of
 * ImageRetrievalNet.forward, networks/imageretrievalnet.py:113):
 *   pooled[n][c] = (mean_{hw} max(x, eps_gem)^p)^(1/p);   out[n][:] = pooled[n][:] / (||pooled[n]||_2 + eps_l2)
 * fmap [n][d][h][w]; pooled, out: [n][d] (the memory the reference's D x N view aliases).  Stand-alone form of the GeM / L2N ops
 * that gdt_net_gem_l2n fuses behind a trunk (any d; SURVEY.md section 8b lists it in the minimum C ABI). */
int gdt_gem_l2n(const float* fmap, int n, int d, int h, int w, float p, float eps_gem, float eps_l2, float* pooled, float* out, void* stream);

/* x / (||x||_2 + eps) over rows of a [N][D] matrix (cirtorch layers/functional.py:130-131) */
int gdt_l2n_rows(const float* x, float* y, int n, int d, float eps, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Retrieval scoring ("next" row of SURVEY.md section 8f: the consumer of the gathered descriptors)
 * Replaces  scores = np.dot(vecs.T, qvecs); ranks = np.argsort(-scores, axis=0)
 *   mdir/components/optim/score/cirscore.py:71-73; mdir/external/cirtorch/datasets/traindataset.py:246-279 (mm + sort).
 * vecs [ndb][d] and qvecs [nq][d]: row-major fp32 descriptor blocks (d a power of two).  scores_t [nq][ndb] is the transpose of the
 * reference's ndb x nq matrix; ranks_t int32 [nq][ndb] holds index_base + database index in order of decreasing score
 * (NULL: scores only).  The GEMM runs in the f16x3 arithmetic (fp32-class accuracy).
 * ------------------------------------------------------------------------------------------------------------------ */
int gdt_retrieval_workspace_bytes(int ndb, int nq, int d, int with_ranks, size_t* bytes);
int gdt_retrieval_scores_ranks(const float* vecs, const float* qvecs, float* scores_t, int* ranks_t, int ndb, int nq, int d,
                               int index_base, void* workspace, size_t workspace_bytes, void* stream);
/* Cluster-aware hard-negative selection on the ranks above -- replaces the Python loop of
 *   TuplesDataset._search_hard_negatives   mdir/external/cirtorch/datasets/traindataset.py:256-275
 * ranks_t [nq][ndb] as written by gdt_retrieval_scores_ranks (pool positions + index_base, best score first), pool_cluster [ndb] / query_cluster [nq]:
 * the cluster id of every pool image / query (self.clusters[...]).  Per query the first `nnum` pool images (1 <= nnum <= 64) whose cluster is neither
 * the query's nor that of an image already taken: neg_pos [nq][nnum] (pool positions + index_base, -1 where the pool ran out of clusters: *status
 * then has bit 0 set -- the reference's loop would raise IndexError), neg_dist [nq][nnum] = ||q - p + 1e-6||_2, the reference's statistic.
 * vecs [ndb][d], qvecs [nq][d] fp32 rows; every pointer is a device buffer. */
int gdt_retrieval_select_negatives(const int* ranks_t, const int* pool_cluster, const int* query_cluster, const float* vecs, const float* qvecs,
                                   int* neg_pos, float* neg_dist, int* status, int ndb, int nq, int d, int nnum, int index_base, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * CLAHE post-processing ("next" row of SURVEY.md section 8f, rank 1: the step between generator and embedder)
 * Replaces the per-image device -> CPU -> cv2 -> device round trip of
 *   ClahePost.postprocess   mdir/components/data/wrapper.py:325-348
 *   ImageClahe.apply        mdir/components/data/transform/functional.py:81-85,147-158 (colorspace "lab": :28-36, :55-63)
 * gdt_clahe_u8: cv2.createCLAHE(clip_limit, (tiles_x, tiles_y)).apply on n uint8 planes [n][h][w] (device buffers).
 * gdt_clahe_lab_f32: x, y fp32 [n][3][h][w] RGB planes (device); rgb = x * in_scale + in_shift (host float[3], NULL = identity),
 *   RGB -> Lab, L quantised to 8 bits, CLAHE, Lab -> RGB, y = (rgb - out_mean) / out_std (host float[3], NULL = identity).
 *   ClahePost(meanstd) is in_scale = out_std = std, in_shift = out_mean = mean.
 * workspace: device scratch of gdt_clahe_workspace_bytes (tile histograms, lookup tables, the 8-bit lightness plane).
 * ------------------------------------------------------------------------------------------------------------------ */
int gdt_clahe_workspace_bytes(int n, int h, int w, int tiles_x, int tiles_y, size_t* bytes);
int gdt_clahe_u8(const unsigned char* src, unsigned char* dst, int n, int h, int w, double clip_limit, int tiles_x, int tiles_y,
                 void* workspace, size_t workspace_bytes, void* stream);
int gdt_clahe_lab_f32(const float* x, float* y, int n, int h, int w, const float* in_scale, const float* in_shift, const float* out_mean,
                      const float* out_std, double clip_limit, int tiles_x, int tiles_y, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Image ingest after decoding ("next" row of SURVEY.md section 8f, rank 3)
 * Replaces  img.thumbnail((s, s), LANCZOS)  (mdir/external/cirtorch/datasets/datahelpers.py:75-82, genericdataset.py:66-102)
 * and  pil2np | totensor | normalize  (mdir/components/data/transform/core_transforms.py:35-100) for a decoded image that
 * already sits in device memory: src [h][w][c] uint8 interleaved (c = 1..4).  Steps, identical to Pillow's integer arithmetic:
 * Image.reduce((fx, fy)) when fx or fy > 1, then Image.resize((out_w, out_h), LANCZOS, box) with `box` (4 host floats, in
 * pixels of the REDUCED image; NULL = all of it).  The caller supplies the plan Pillow's Python layer computes (thumbnail size,
 * reducing_gap factors, box): gandtr_amd/ingest.py mirrors it.  Outputs (either may be NULL): dst_hwc [out_h][out_w][c] uint8;
 * dst_chw [c][out_h][out_w] fp32 = (v / 255 - mean[c]) / std[c]  (host float[c]; NULL = 0 / 1).
 * ------------------------------------------------------------------------------------------------------------------ */
int gdt_ingest_workspace_bytes(int h, int w, int c, int fx, int fy, int out_w, int out_h, size_t* bytes);
int gdt_ingest_resize_u8(const unsigned char* src, int h, int w, int c, int fx, int fy, const float* box, int out_w, int out_h,
                         unsigned char* dst_hwc, float* dst_chw, const float* mean, const float* std, void* workspace, size_t workspace_bytes,
                         void* stream);

/* The same for a LIST of decoded images of different sizes in one call (the reference is batch-1 over exactly such lists:
 * mdir/external/cirtorch/networks/imageretrievalnet.py:319-322, genericdataset.py:66-102).  Every item carries what
 * gdt_ingest_resize_u8 takes per image (box all zero = the whole reduced image).  RGB images on 4-byte-aligned buffers run as
 * three launches for the whole list (box reduction, horizontal pass, vertical pass + conversion; blockIdx.z = image) driven by a
 * descriptor array the call uploads into the workspace; other channel counts / alignments run image by image inside the call.
 * Results are bit-identical to the per-image entry point.  items: host array. */
typedef struct gdt_ingest_item {
    const unsigned char* src;      /* device, [h][w][c] uint8 */
    int h, w, fx, fy;
    float box[4];
    int out_w, out_h;
    unsigned char* dst_hwc;        /* device [out_h][out_w][c] uint8, or NULL */
    float* dst_chw;                /* device [c][out_h][out_w] fp32, or NULL */
} gdt_ingest_item;
int gdt_ingest_batch_workspace_bytes(const gdt_ingest_item* items, int n, int c, size_t* bytes);
int gdt_ingest_resize_u8_batch(const gdt_ingest_item* items, int n, int c, const float* mean, const float* std, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * JPEG decoding on the device (the first half of the ingest row)
 * Replaces  pil_loader: Image.open(f).convert('RGB')  (mdir/external/cirtorch/datasets/datahelpers.py:39-47, called per image from
 * genericdataset.py:66-102) for baseline JPEG files: 8-bit, Huffman-coded, one interleaved scan, grayscale or YCbCr with 4:4:4 / 4:2:2 /
 * 4:2:0 sampling, with or without restart markers.  The output is bit-identical to Pillow's (libjpeg-turbo defaults: the integer
 * "islow" inverse DCT, "fancy" triangle chroma upsampling, the fixed-point YCbCr -> RGB tables).  Anything else (progressive,
 * arithmetic-coded, 12-bit, CMYK / RGB-coded files, other sampling factors) is reported as GDT_ERR_INVALID by gdt_jpeg_parse so that the
 * caller can hand that file to its general-purpose loader; nothing is decoded on the host here.
 *
 * Host side (pure C, no device): gdt_jpeg_parse reads the headers of one file into a gdt_jpeg_info (geometry, tables, where the
 * entropy-coded segment lies, how many restart intervals it has); gdt_jpeg_extract_scan copies that segment with the 0xFF00 byte
 * stuffing and the restart markers removed into `dst` (capacity info.scan_capacity; to be uploaded) and writes the info.nsegments + 1
 * byte offsets of the restart intervals within it.
 * Device side: gdt_jpeg_decode_u8_batch decodes a list of such images in a fixed number of launches for the whole list.  Huffman
 * decoding is parallel WITHIN a file: every 128-byte piece of an interval is decoded by its own thread from a guessed state, the
 * guesses are repaired by re-decoding from the neighbour's exit state until nothing changes (Huffman streams self-synchronise within a
 * few symbols; the call reads one flag per round of four launches, so it synchronises the stream), then the coefficients are written,
 * the DC predictions are prefix-summed per interval and component, and dequantisation + inverse DCT, upsampling and colour conversion
 * run per block / per pixel.  mode = 1 decodes every interval with ONE thread instead (the checker of the parallel decoder).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct gdt_jpeg_info {
    int width, height, ncomp;                 /* ncomp 1 (grayscale), 3 (YCbCr) or 4 (CMYK / YCCK, see adobe_transform) */
    int hs[4], vs[4], tq[4], td[4], ta[4];     /* per component: sampling factors, quantisation / DC / AC table numbers */
    int restart_interval;                      /* in MCUs, 0 = none */
    int mcus_x, mcus_y, blocks_per_mcu;
    int nsegments;                             /* restart intervals in the scan (1 without restart markers) */
    unsigned long long scan_offset;            /* first entropy-coded byte in the file */
    unsigned long long scan_capacity;          /* bytes gdt_jpeg_extract_scan may write (>= the unstuffed size + padding) */
    unsigned short quant[4][64];               /* natural (row-major) order */
    unsigned char huff_bits[4][17];            /* tables 0-1: DC 0 / 1, 2-3: AC 0 / 1; bits[l] = number of codes of length l */
    unsigned char huff_vals[4][256];
    int progressive;                           /* 1: SOF2 (progressive DCT, Huffman): scan_offset = the first SOS marker; decoded through
                                                  gdt_jpeg_progressive_coefficients + gdt_jpeg_decode_coef_u8_batch (below) */
    int comp_id[4];                            /* progressive files: the frame's component identifiers (their scans name components by id) */
    int adobe_transform;                       /* four-component files: 2 = YCCK (Adobe APP14 transform 2), 0 = CMYK (transform 0, or no Adobe marker); the decoded
                                                  CMYK samples are taken as inverted ("Adobe convention", as Pillow's JPEG plugin does for every CMYK file) and
                                                  converted to RGB with Pillow's convert('RGB') arithmetic: the reference's pil_loader, datahelpers.py:39-47 */
} gdt_jpeg_info;
int gdt_jpeg_parse(const unsigned char* file, size_t nbytes, gdt_jpeg_info* info);
int gdt_jpeg_extract_scan(const unsigned char* file, size_t nbytes, const gdt_jpeg_info* info, unsigned char* dst, unsigned int* seg_off);
/* the same two steps for a list of files on `threads` host threads (1 = in the calling thread): status[i] = what gdt_jpeg_parse would return for file i
 * (call it on that file for the message); file i's scan goes to dst + dst_off[i] (info.scan_capacity bytes) and its nsegments + 1 offsets to
 * seg_off + seg_index[i]. */
int gdt_jpeg_parse_batch(const unsigned char* const* files, const size_t* nbytes, int n, gdt_jpeg_info* infos, int* status, int threads);
int gdt_jpeg_extract_scan_batch(const unsigned char* const* files, const size_t* nbytes, const gdt_jpeg_info* infos, int n, unsigned char* dst,
                                const size_t* dst_off, unsigned int* seg_off, const size_t* seg_index, int threads);
typedef struct gdt_jpeg_item {
    const gdt_jpeg_info* info;     /* host */
    const unsigned char* scan;     /* device: the bytes gdt_jpeg_extract_scan produced (scan_capacity of them) */
    const unsigned int* seg_off;   /* host: info->nsegments + 1 offsets into scan */
    unsigned char* dst_hwc;        /* device: [height][width][3] uint8 RGB (grayscale replicated, CMYK converted, as convert('RGB') does) */
} gdt_jpeg_item;
int gdt_jpeg_decode_workspace_bytes(const gdt_jpeg_item* items, int n, size_t* bytes);
int gdt_jpeg_decode_u8_batch(const gdt_jpeg_item* items, int n, int mode, void* workspace, size_t workspace_bytes, void* stream);
/* Progressive files (SOF2: spectral selection + successive approximation, Huffman-coded; what pil_loader opens just as well,
 * datahelpers.py:39-47).  Their scans refine the SAME coefficients several times over, so the entropy decoding is a sequential pass per
 * scan: gdt_jpeg_progressive_coefficients (host, pure C; ITU T.81 Annex G) walks every scan of one file and writes the quantised
 * coefficients, DC prediction resolved, in the block order of the device pipeline (MCU by MCU, luma blocks first), 64 natural-order
 * int16 per block: info->mcus_x * mcus_y * blocks_per_mcu blocks.  gdt_jpeg_progressive_coefficients_batch: the same for a list on a few
 * host threads, file i at coef + coef_off[i] elements.  gdt_jpeg_decode_coef_u8_batch (device): dequantisation + inverse DCT, upsampling,
 * colour conversion of n images whose coefficients lie at coef_dev + coef_off[i] (one device buffer): the back half of
 * gdt_jpeg_decode_u8_batch, same arithmetic, same [height][width][3] uint8 output.  Baseline files may be decoded this way too. */
int gdt_jpeg_progressive_coefficients(const unsigned char* file, size_t nbytes, const gdt_jpeg_info* info, short* coef);
int gdt_jpeg_progressive_coefficients_batch(const unsigned char* const* files, const size_t* nbytes, const gdt_jpeg_info* infos, int n, short* coef,
                                            const size_t* coef_off, int* status, int threads);
int gdt_jpeg_decode_coef_workspace_bytes(const gdt_jpeg_info* infos, int n, size_t* bytes);
int gdt_jpeg_decode_coef_u8_batch(const gdt_jpeg_info* infos, const short* coef_dev, const size_t* coef_off, unsigned char* const* dst_hwc, int n,
                                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Learned whitening ("next" row of SURVEY.md section 8f, rank 4): the {m, P} that gdt_whiten applies
 * Replaces  whitenlearn(X, qidxs, pidxs)  (mdir/external/cirtorch/utils/whiten.py:37-70; called with float64 D x N values from
 * mdir/stages/whiten.py:30-75).  x: [n_vec][d] fp32 descriptor rows (device), qidx / pidx: int32 [n_pairs] (device) row numbers of
 * the matching query / positive pairs.  All arithmetic in float64.  Outputs (device): m [d], P [d][d] row-major, eig [d] (may be
 * NULL) = eigenvalues in decreasing order.  info (host, may be NULL): info[0] = diagonal-jitter steps of the Cholesky
 * (whiten.py:55-70), info[1] = Jacobi sweeps.  The call synchronises the stream (convergence checks).  Rows of P are defined up
 * to sign (eigenvectors).  Returns GDT_ERR_NOT_CONVERGED (4) when the eigensolver stops at its sweep cap (60 one-sided / 40 two-sided
 * sweeps) without meeting its convergence test; the outputs are then best-effort values.
 * ------------------------------------------------------------------------------------------------------------------ */
int gdt_whiten_learn_workspace_bytes(int n_vec, int d, int n_pairs, size_t* bytes);
int gdt_whiten_learn(const float* x, const int* qidx, const int* pidx, int n_vec, int d, int n_pairs, double* m_out, double* p_out,
                     double* eig_out, int* info, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): sustained rate of the matrix pipe alone on this device -- a kernel of nothing
 * but v_mfma_f32_32x32x16_f16 on random fp16 operands (8 independent accumulators per wave, 8 waves per CU) run for about
 * `millis` milliseconds.  bench.py reports it next to the 2.5 PFLOP/s datasheet peak: these boxes throttle to 1.5-1.7 PFLOP/s
 * under sustained matrix load (profiles/experiments/mfma_peak.hip is the stand-alone version).
 * ------------------------------------------------------------------------------------------------------------------ */
int gdt_mfma_only_tflops(int millis, double* tflops, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GANDTR_HIP_H */
