// Retrieval scoring on the device: the immediate consumer of the all-gathered descriptors.
// Replaces   scores = np.dot(vecs.T, qvecs); ranks = np.argsort(-scores, axis=0)
//   mdir/components/optim/score/cirscore.py:71-73 (evaluation), mdir/external/cirtorch/datasets/traindataset.py:246-279
//   (hard-negative mining: torch.mm + torch.sort).
// scores: one f16x3 GEMM (split-fp16, fp32-class accuracy -- the ranking of near-ties must not depend on 11-bit operands)
// run by the 1x1 path of conv_igemm_x3.hip with M = database size, K = D, N = queries, written directly as [nq][ndb];
// ranks: rocPRIM segmented radix sort (descending) of (score, database index) pairs, one segment per query.
#include <algorithm>
#include <cstring>
#include <string.h>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "../../include/gandtr_hip.h"
#include "gdt_common.h"

namespace {

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

// q [nq][d] fp32 -> hi / lo fp16 [nq_pad][kpad], zero padded
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ q, f16* __restrict__ hi, f16* __restrict__ lo, int nq,
                                                         int d, int nq_pad, int kpad) {
    const long total = (long)nq_pad * kpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / kpad), c = (int)(i % kpad);
        const float x = (r < nq && c < d) ? q[(long)r * d + c] : 0.f;
        const f16 h = (f16)x;
        hi[i] = h;
        lo[i] = (f16)((x - (float)h) * 2048.f);
    }
}

__global__ __launch_bounds__(256) void iota_segments_kernel(int* __restrict__ idx, int* __restrict__ offsets, int nseg, int len, int base) {
    const long total = (long)nseg * len;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) idx[i] = base + (int)(i % len);
    if (blockIdx.x == 0)
        for (int s = threadIdx.x; s <= nseg; s += 256) offsets[s] = s * len;
}

struct Layout { size_t hi, lo, idx, keys, offs, tmp, tmp_bytes, total; int nq_pad, kpad; };

int plan(int ndb, int nq, int d, bool ranks, Layout& L) {
    GDT_REQUIRE(ndb >= 1 && nq >= 1 && d >= 8 && (d & (d - 1)) == 0, "retrieval needs ndb, nq >= 1 and a power-of-two descriptor size >= 8");
    GDT_REQUIRE((long)ndb * nq < (1l << 31), "ndb * nq must stay below 2^31");
    const int bn = gdt_conv_bn(nq);
    L.nq_pad = (nq + bn - 1) / bn * bn;
    L.kpad = (d + 63) / 64 * 64;
    size_t off = 0;
    L.hi = off; off += align_up((size_t)L.nq_pad * L.kpad * sizeof(f16));
    L.lo = off; off += align_up((size_t)L.nq_pad * L.kpad * sizeof(f16));
    L.idx = L.keys = L.offs = L.tmp = off; L.tmp_bytes = 0;
    if (ranks) {
        L.idx = off; off += align_up((size_t)nq * ndb * sizeof(int));
        L.keys = off; off += align_up((size_t)nq * ndb * sizeof(float));
        L.offs = off; off += align_up((size_t)(nq + 1) * sizeof(int));
        size_t tb = 0;
        hipError_t e = rocprim::segmented_radix_sort_pairs_desc(nullptr, tb, (const float*)nullptr, (float*)nullptr, (const int*)nullptr,
                                                                 (int*)nullptr, (unsigned)((size_t)nq * ndb), (unsigned)nq,
                                                                 (const int*)nullptr, (const int*)nullptr, 0, 32, (hipStream_t)0);
        GDT_CHECK_HIP(e);
        L.tmp = off; L.tmp_bytes = tb; off += align_up(tb);
    }
    L.total = off + ALIGN;
    return GDT_OK;
}


// Cluster-aware hard-negative selection (mdir/external/cirtorch/datasets/traindataset.py:256-275): per query walk its ranked pool from the
// best score down and take the first `nnum` images that belong neither to the query's cluster nor to the cluster of an image already taken;
// the statistic of the reference, ||q - p + 1e-6||_2 per chosen negative, is evaluated by the whole wave.  One wave per query: lane 0 walks
// (a dependent chain of two small loads per step), all lanes compute the distances.
__global__ __launch_bounds__(64) void select_negatives_kernel(const int* __restrict__ ranks_t, const int* __restrict__ pool_cluster,
                                                              const int* __restrict__ query_cluster, const float* __restrict__ vecs,
                                                              const float* __restrict__ qvecs, int* __restrict__ neg_pos, float* __restrict__ neg_dist,
                                                              int* __restrict__ status, int ndb, int nq, int d, int nnum, int index_base) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= nq) return;
    __shared__ int chosen[64];
    if (lane == 0) {
        int used[65];
        int nused = 1, n = 0;
        used[0] = query_cluster[q];
        const int* rk = ranks_t + (size_t)q * ndb;
        for (int r = 0; r < ndb && n < nnum; ++r) {
            const int p = rk[r] - index_base;
            const int c = pool_cluster[p];
            bool seen = false;
            for (int k = 0; k < nused; ++k) seen |= used[k] == c;
            if (!seen) { chosen[n++] = p; used[nused++] = c; }
        }
        for (int k = n; k < nnum; ++k) chosen[k] = -1;
        if (n < nnum) atomicOr(status, 1);          // the pool ran out of clusters: the reference's loop would index past the ranks
    }
    __syncthreads();
    for (int k = 0; k < nnum; ++k) {
        const int p = chosen[k];
        float acc = 0.f;
        if (p >= 0)
            for (int i = lane; i < d; i += 64) {
                const float df = qvecs[(size_t)q * d + i] - vecs[(size_t)p * d + i] + 1e-6f;
                acc += df * df;
            }
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
        if (lane == 0) { neg_pos[(size_t)q * nnum + k] = p < 0 ? -1 : p + index_base; neg_dist[(size_t)q * nnum + k] = p < 0 ? 0.f : sqrtf(acc); }
    }
}

}  // namespace

extern "C" {

int gdt_retrieval_workspace_bytes(int ndb, int nq, int d, int with_ranks, size_t* bytes) {
    GDT_REQUIRE(bytes != nullptr, "bytes");
    Layout L;
    int rc = plan(ndb, nq, d, with_ranks != 0, L);
    if (rc != GDT_OK) return rc;
    *bytes = L.total;
    return GDT_OK;
}

int gdt_retrieval_scores_ranks(const float* vecs, const float* qvecs, float* scores_t, int* ranks_t, int ndb, int nq, int d,
                               int index_base, void* workspace, size_t workspace_bytes, void* stream) {
    GDT_REQUIRE(vecs && qvecs && scores_t && workspace, "null buffer");
    Layout L;
    int rc = plan(ndb, nq, d, ranks_t != nullptr, L);
    if (rc != GDT_OK) return rc;
    if (L.total > workspace_bytes) {
        gdt_set_error("workspace too small: need " + std::to_string(L.total) + " bytes, got " + std::to_string(workspace_bytes));
        return GDT_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)(((uintptr_t)workspace + ALIGN - 1) / ALIGN * ALIGN);
    f16* hi = (f16*)(ws + L.hi);
    f16* lo = (f16*)(ws + L.lo);
    {
        const long total = (long)L.nq_pad * L.kpad;
        const int grid = (int)std::min<long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(split_rows_kernel, dim3(grid), dim3(256), 0, st, qvecs, hi, lo, nq, d, L.nq_pad, L.kpad);
        GDT_CHECK_HIP(hipGetLastError());
    }
    // scores_t[q][i] = <vecs[i], qvecs[q]>: a 1x1 "convolution" over ndb positions with d input and nq output channels
    ConvLaunch c{};
    c.in = (const f16*)vecs; c.w = hi; c.w_lo = lo; c.out_f32 = scores_t; c.zeros = hi;
    c.N = 1; c.H = 1; c.W = ndb; c.Cin = d; c.lc8 = 0; while ((8 << c.lc8) < d) ++c.lc8;
    c.Cout = nq; c.CoutPad = L.nq_pad; c.Kpad = L.kpad; c.nk = L.kpad / 32;
    c.OHg = 1; c.OWg = ndb; c.OH = 1; c.OW = ndb; c.sy = c.sx = 1; c.osy = c.osx = 1;
    c.ntaps = 1; c.TW = 1; c.invTW = 65536; c.dys = c.dxs = 1;
    c.M = ndb;
    rc = gdt_launch_conv_x3(c, st, nullptr);
    if (rc != GDT_OK || !ranks_t) return rc;
    int* idx = (int*)(ws + L.idx);
    int* offs = (int*)(ws + L.offs);
    {
        const long total = (long)nq * ndb;
        const int grid = (int)std::min<long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(iota_segments_kernel, dim3(grid), dim3(256), 0, st, idx, offs, nq, ndb, index_base);
        GDT_CHECK_HIP(hipGetLastError());
    }
    size_t tb = L.tmp_bytes;
    GDT_CHECK_HIP(rocprim::segmented_radix_sort_pairs_desc((void*)(ws + L.tmp), tb, (const float*)scores_t, (float*)(ws + L.keys),
                                                            (const int*)idx, ranks_t, (unsigned)((size_t)nq * ndb), (unsigned)nq,
                                                            (const int*)offs, (const int*)(offs + 1), 0, 32, st));
    return GDT_OK;
}

int gdt_retrieval_select_negatives(const int* ranks_t, const int* pool_cluster, const int* query_cluster, const float* vecs, const float* qvecs,
                                   int* neg_pos, float* neg_dist, int* status, int ndb, int nq, int d, int nnum, int index_base, void* stream) {
    GDT_REQUIRE(ranks_t && pool_cluster && query_cluster && vecs && qvecs && neg_pos && neg_dist && status, "null buffer");
    GDT_REQUIRE(ndb >= 1 && nq >= 1 && d >= 1 && nnum >= 1 && nnum <= 64, "1 <= nnum <= 64 negatives per query");
    hipStream_t st = (hipStream_t)stream;
    GDT_CHECK_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    hipLaunchKernelGGL(select_negatives_kernel, dim3(nq), dim3(64), 0, st, ranks_t, pool_cluster, query_cluster, vecs, qvecs, neg_pos, neg_dist, status,
                       ndb, nq, d, nnum, index_base);
    GDT_CHECK_HIP(hipGetLastError());
    return GDT_OK;
}

}  // extern "C"
