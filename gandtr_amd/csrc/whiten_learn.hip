// Learned whitening ("Lw") on the device, in float64: produces the {m, P} that CirtorchWhiten consumes.
//   whitenlearn / cholesky      mdir/external/cirtorch/utils/whiten.py:37-70
//   learn_lw_whitening          mdir/stages/whiten.py:30-75 (values.astype(float64).T, query / positive index lists)
// Reference (numpy, float64):  m = mean(X[:, q]);  S = df df^T / n  with  df = X[:, q] - X[:, p];  P0 = inv(cholesky(S))  (with
// a growing diagonal jitter until S is positive definite);  D = (P0 (X - m)) (P0 (X - m))^T;  eigen-decomposition of D, eigenvalues
// in decreasing order;  P = eigvec^T P0.
// Here:  the two O(d^2 N) covariance products are one tiled float64 GEMM kernel reading the float32 descriptor rows directly
// (pair differences / mean-centred rows formed on the fly, fixed summation order => deterministic and exactly symmetric);
// with C = sum (x - m)(x - m)^T the reference's d x N product is never materialised: the matrix to decompose is P0 C P0^T = G G^T for
// G = P0 chol(C), and its eigenpairs are the left singular vectors / squared singular values of G, found by a one-sided (Hestenes)
// Jacobi with both columns of a pair in LDS (one pass over the matrix per round, HBM-bound at 5.9 TB/s); a two-sided cyclic Jacobi
// on P0 C P0^T is the fallback when C is singular.  Cholesky and the triangular inverse run as d short launches each (d <= 2048: a
// few tens of ms).  Eigenvectors are defined up to sign, so P's rows are too.
#include <math.h>
#include <string.h>

#include "../../include/gandtr_hip.h"
#include "gdt_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------- GEMM
// C[m][n] = scale * sum_k A(k, m) * B(k, n)   (both operands K-major), 64 x 64 tile per workgroup, 4 x 4 per lane, K-step 16.
// Operand sources: MAT  = double matrix [K][ld];  DIFF = float rows X[q[k]] - X[p[k]];  CENT = float row X[k] - mean.
enum { SRC_MAT = 0, SRC_DIFF = 1, SRC_CENT = 2 };
struct Operand {
    int kind;
    const double* mat; int ld;          // SRC_MAT
    const float* x; int xd;             // SRC_DIFF / SRC_CENT: rows of xd floats
    const int* q; const int* p;         // SRC_DIFF
    const double* mean;                 // SRC_CENT
};

__device__ __forceinline__ double fetch(const Operand& o, int k, int m) {
    if (o.kind == SRC_MAT) return o.mat[(size_t)k * o.ld + m];
    if (o.kind == SRC_DIFF) return (double)o.x[(size_t)o.q[k] * o.xd + m] - (double)o.x[(size_t)o.p[k] * o.xd + m];
    return (double)o.x[(size_t)k * o.xd + m] - o.mean[m];
}

__global__ __launch_bounds__(256) void gemm_kk_kernel(Operand a, Operand b, double* __restrict__ c, int M, int N, int K, double scale) {
    __shared__ double as[16][64 + 1];
    __shared__ double bs[16][64 + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                       // 16 x 64 elements per operand, 4 per lane
            const int e = tid + i * 256, kk = e >> 6, mm = e & 63;
            const bool ok = k0 + kk < K;
            as[kk][mm] = (ok && m0 + mm < M) ? fetch(a, k0 + kk, m0 + mm) : 0.0;
            bs[kk][mm] = (ok && n0 + mm < N) ? fetch(b, k0 + kk, n0 + mm) : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = as[kk][ty * 4 + i]; bv[i] = bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < M && n < N) c[(size_t)m * N + n] = acc[i][j] * scale;
        }
}

__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ a, double* __restrict__ t, int d) {
    __shared__ double tile[32][33];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y0 = blockIdx.y * 32;
    for (int r = threadIdx.x >> 5; r < 32; r += 8)
        if (x < d && y0 + r < d) tile[r][threadIdx.x & 31] = a[(size_t)(y0 + r) * d + x];
    __syncthreads();
    const int xt = blockIdx.y * 32 + (threadIdx.x & 31), yt0 = blockIdx.x * 32;
    for (int r = threadIdx.x >> 5; r < 32; r += 8)
        if (xt < d && yt0 + r < d) t[(size_t)(yt0 + r) * d + xt] = tile[threadIdx.x & 31][r];
}

// m[c] = mean over the query rows, sequential in pair order (one lane per column)
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, const int* __restrict__ q, int n, int d, double* __restrict__ m) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)x[(size_t)q[i] * d + c];
    m[c] = s / (double)n;
}

// ---------------------------------------------------------------------------------------------------------------- Cholesky
// a: working copy of S (+ alpha I), overwritten by L in its lower triangle.  Right-looking, one column per step.
__global__ __launch_bounds__(256) void chol_column_kernel(double* __restrict__ a, int d, int k, int* __restrict__ fail) {
    const double piv = a[(size_t)k * d + k];           // stays in place until chol_diag_kernel: every workgroup of this launch reads it
    if (!(piv > 0.0)) { if (threadIdx.x == 0 && blockIdx.x == 0) *fail = 1; return; }
    const double l = sqrt(piv);
    const int i = k + 1 + blockIdx.x * 256 + threadIdx.x;
    if (i < d) a[(size_t)i * d + k] /= l;
}
__global__ __launch_bounds__(256) void chol_diag_kernel(double* __restrict__ a, int d) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < d) a[(size_t)k * d + k] = sqrt(a[(size_t)k * d + k]);
}
// a <- (a + a^T) / 2 (the two triple products are equal only up to rounding)
__global__ __launch_bounds__(256) void symmetrise_kernel(double* __restrict__ a, int d) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)d * d) return;
    const int r = (int)(i / d), c = (int)(i % d);
    if (c <= r) return;
    const double v = 0.5 * (a[(size_t)r * d + c] + a[(size_t)c * d + r]);
    a[(size_t)r * d + c] = v; a[(size_t)c * d + r] = v;
}
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ a, int d, int k, const int* __restrict__ fail) {
    if (*fail) return;
    const int j = k + 1 + blockIdx.x * 64 + (threadIdx.x & 63), i = k + 1 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= d || j > i) return;                                       // lower triangle only
    a[(size_t)i * d + j] -= a[(size_t)i * d + k] * a[(size_t)j * d + k];
}
__global__ __launch_bounds__(256) void add_diag_copy_kernel(const double* __restrict__ s, double* __restrict__ a, int d, double alpha) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)d * d) return;
    const int r = (int)(i / d), c = (int)(i % d);
    a[i] = s[i] + (r == c ? alpha : 0.0);
}
// X = L^-1 (lower triangular), one row per step: X[k][j] = (delta_kj - sum_{j<=t<k} L[k][t] X[t][j]) / L[k][k].  The sum over t is
// split into TRI_PARTS fixed chunks (one workgroup each, partial sums in `part`), combined in chunk order by the second kernel.
constexpr int TRI_PARTS = 16;
__global__ __launch_bounds__(256) void tri_inverse_partial_kernel(const double* __restrict__ l, const double* __restrict__ x,
                                                                  double* __restrict__ part, int d, int k) {
    const int j = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    if (j >= d) return;
    const int len = (k + TRI_PARTS - 1) / TRI_PARTS, t0 = max(c * len, j), t1 = min((c + 1) * len, k);     // X[t][j] = 0 for t < j
    double s = 0.0;
    for (int t = t0; t < t1; ++t) s += l[(size_t)k * d + t] * x[(size_t)t * d + j];
    part[(size_t)c * d + j] = s;
}
__global__ __launch_bounds__(256) void tri_inverse_finish_kernel(const double* __restrict__ l, double* __restrict__ x,
                                                                 const double* __restrict__ part, int d, int k) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    double s = 0.0;
    if (j <= k) {
        s = (j == k) ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < TRI_PARTS; ++c) s -= part[(size_t)c * d + j];
        s /= l[(size_t)k * d + k];
    }
    x[(size_t)k * d + j] = s;
}

// ---------------------------------------------------------------------------------------------------------------- Jacobi
// round-robin tournament on d players (d even): round r, pair k -> (p, q), p < q
__device__ __forceinline__ void pair_of(int d, int r, int k, int& p, int& q) {
    const int n1 = d - 1;
    int a, b;
    if (k == 0) { a = n1; b = r % n1; }
    else { a = (r + k) % n1; b = (r - k + n1) % n1; }
    p = min(a, b); q = max(a, b);
}
__global__ __launch_bounds__(256) void jacobi_angles_kernel(const double* __restrict__ a, int d, int r, double* __restrict__ cs) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= d / 2) return;
    int p, q;
    pair_of(d, r, k, p, q);
    const double apq = a[(size_t)p * d + q], app = a[(size_t)p * d + p], aqq = a[(size_t)q * d + q];
    double c = 1.0, s = 0.0;
    if (fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * sqrt(fabs(app * aqq))) {
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        c = 1.0 / sqrt(1.0 + t * t);
        s = t * c;
    }
    cs[2 * k] = c; cs[2 * k + 1] = s;
}
// A <- J^T A: rows p and q of every pair, coalesced over the columns
__global__ __launch_bounds__(256) void jacobi_rows_kernel(double* __restrict__ a, int d, int r, const double* __restrict__ cs) {
    const int k = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= d) return;
    int p, q;
    pair_of(d, r, k, p, q);
    const double c = cs[2 * k], s = cs[2 * k + 1];
    const double ap = a[(size_t)p * d + j], aq = a[(size_t)q * d + j];
    a[(size_t)p * d + j] = c * ap - s * aq;
    a[(size_t)q * d + j] = s * ap + c * aq;
}
// M <- M J for M = A (rows 0..d-1) and V (rows d..2d-1): one workgroup per row, the row held in LDS
__global__ __launch_bounds__(256) void jacobi_cols_kernel(double* __restrict__ a, double* __restrict__ v, int d, int r,
                                                          const double* __restrict__ cs) {
    extern __shared__ double row[];
    double* m = (blockIdx.x < (unsigned)d) ? a + (size_t)blockIdx.x * d : v + (size_t)(blockIdx.x - d) * d;
    for (int j = threadIdx.x; j < d; j += 256) row[j] = m[j];
    __syncthreads();
    for (int k = threadIdx.x; k < d / 2; k += 256) {
        int p, q;
        pair_of(d, r, k, p, q);
        const double c = cs[2 * k], s = cs[2 * k + 1];
        const double mp = row[p], mq = row[q];
        row[p] = c * mp - s * mq;
        row[q] = s * mp + c * mq;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < d; j += 256) m[j] = row[j];
}
// out[0] = sum of squared off-diagonal entries, out[1] = sum of squared diagonal entries (fixed order: one workgroup)
__global__ __launch_bounds__(1024) void offdiag_kernel(const double* __restrict__ a, int d, double* __restrict__ out) {
    __shared__ double so[1024], sd[1024];
    double o = 0.0, g = 0.0;
    for (size_t i = threadIdx.x; i < (size_t)d * d; i += 1024) {
        const double v = a[i];
        if (i / d == i % d) g += v * v; else o += v * v;
    }
    so[threadIdx.x] = o; sd[threadIdx.x] = g;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { so[threadIdx.x] += so[threadIdx.x + s]; sd[threadIdx.x] += sd[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = so[0]; out[1] = sd[0]; }
}
__global__ __launch_bounds__(256) void identity_kernel(double* __restrict__ v, int d) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)d * d) v[i] = (i / d == i % d) ? 1.0 : 0.0;
}
// values a[i * stride] (the diagonal of A: stride d + 1); order[i] = index of the i-th largest (ties by index): O(d^2) ranking, d <= a few thousand
__global__ __launch_bounds__(256) void rank_desc_kernel(const double* __restrict__ a, size_t stride, int d, int* __restrict__ order,
                                                        double* __restrict__ eig) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d) return;
    const double li = a[(size_t)i * stride];
    int rank = 0;
    for (int j = 0; j < d; ++j) {
        const double lj = a[(size_t)j * stride];
        rank += (lj > li) || (lj == li && j < i);
    }
    order[rank] = i;
    if (eig) eig[rank] = li;
}
// vs[k][i] = v[k][order[i]]
__global__ __launch_bounds__(256) void gather_cols_kernel(const double* __restrict__ v, const int* __restrict__ order, double* __restrict__ vs, int d) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)d * d) return;
    const int k = (int)(i / d), c = (int)(i % d);
    vs[i] = v[(size_t)k * d + order[c]];
}

// ---------------------------------------------------------------------------------------------------------------- one-sided Jacobi
// With C = Lc Lc^T the matrix D = P0 C P0^T is G G^T for G = P0 Lc, so the eigenvectors of D are the left singular vectors of G and
// the eigenvalues the squared singular values: Hestenes' one-sided Jacobi orthogonalises the COLUMNS of G by plane rotations.  gt
// holds G transposed (row i = column i of G, contiguous).  One workgroup per pair: both columns (2 x d doubles) sit in LDS between
// the dot products and the rotation, so a round moves the matrix once (the two-sided form below moves A and V three times).
__global__ __launch_bounds__(256) void hestenes_round_kernel(double* __restrict__ gt, int d, int r, int* __restrict__ changed, double tol2) {
    extern __shared__ double cols[];                 // [2][d]
    __shared__ double part[3][4];
    int p, q;
    pair_of(d, r, blockIdx.x, p, q);
    double* rp = gt + (size_t)p * d;
    double* rq = gt + (size_t)q * d;
    double a = 0.0, b = 0.0, g = 0.0;
    for (int j = threadIdx.x; j < d; j += 256) {
        const double x = rp[j], y = rq[j];
        cols[j] = x; cols[d + j] = y;
        a = fma(x, x, a); b = fma(y, y, b); g = fma(x, y, g);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); g += __shfl_xor(g, m); }      // fixed butterfly order
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = a; part[1][threadIdx.x >> 6] = b; part[2][threadIdx.x >> 6] = g; }
    __syncthreads();
    a = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
    b = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    g = (part[2][0] + part[2][1]) + (part[2][2] + part[2][3]);
    if (!(g * g > tol2 * a * b)) return;             // already orthogonal to working precision (uniform across the workgroup)
    if (threadIdx.x == 0) *changed = 1;
    const double zeta = (b - a) / (2.0 * g);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
    for (int j = threadIdx.x; j < d; j += 256) {
        const double x = cols[j], y = cols[d + j];
        rp[j] = c * x - s * y;
        rq[j] = s * x + c * y;
    }
}
// lam[i] = |row i|^2 (one workgroup per row, fixed reduction order)
__global__ __launch_bounds__(256) void row_norm2_kernel(const double* __restrict__ gt, int d, double* __restrict__ lam) {
    __shared__ double part[4];
    const double* row = gt + (size_t)blockIdx.x * d;
    double a = 0.0;
    for (int j = threadIdx.x; j < d; j += 256) a = fma(row[j], row[j], a);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) lam[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
// us[k][i] = gt[order[i]][k] / sqrt(lam[order[i]]): unit left singular vectors as columns, in eigenvalue order
__global__ __launch_bounds__(256) void unit_columns_kernel(const double* __restrict__ gt, const double* __restrict__ lam,
                                                           const int* __restrict__ order, double* __restrict__ us, int d) {
    __shared__ double tile[32][33];
    const int i0 = blockIdx.y * 32, k0 = blockIdx.x * 32, tx = threadIdx.x & 31;
    for (int r = threadIdx.x >> 5; r < 32; r += 8) {
        const int i = i0 + r, k = k0 + tx;
        if (i < d && k < d) { const int src = order[i]; tile[r][tx] = gt[(size_t)src * d + k] / sqrt(lam[src]); }
    }
    __syncthreads();
    for (int r = threadIdx.x >> 5; r < 32; r += 8) {
        const int k = k0 + r, i = i0 + tx;
        if (i < d && k < d) us[(size_t)k * d + i] = tile[tx][r];
    }
}
__global__ __launch_bounds__(256) void zero_upper_kernel(double* __restrict__ a, int d) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)d * d && (int)(i % d) > (int)(i / d)) a[i] = 0.0;
}

constexpr size_t ALIGN = 256;
inline size_t align_up(size_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }
struct Layout { size_t s, l, p0, t1, t2, a, v, cs, flag, red, order, total; };
int plan(int n_vec, int d, int n_pairs, Layout& L) {
    GDT_REQUIRE(n_vec >= 1 && n_pairs >= 1, "whiten_learn: empty input");
    GDT_REQUIRE(d >= 2 && d % 2 == 0 && d <= 8192, "whiten_learn: descriptor size must be even and <= 8192");
    const size_t dd = align_up((size_t)d * d * sizeof(double));
    size_t off = 0;
    L.s = off; off += dd; L.l = off; off += dd; L.p0 = off; off += dd; L.t1 = off; off += dd; L.t2 = off; off += dd;
    L.a = off; off += dd; L.v = off; off += dd;
    L.cs = off; off += align_up((size_t)d * sizeof(double));
    L.flag = off; off += ALIGN; L.red = off; off += ALIGN;
    L.order = off; off += align_up((size_t)d * sizeof(int));
    L.total = off + ALIGN;
    return GDT_OK;
}

// in-place right-looking Cholesky of the lower triangle of `a`; failed = 1 when a pivot is not positive (synchronises the stream)
int cholesky_inplace(hipStream_t st, double* a, int d, int* flag, int& failed) {
    GDT_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
    for (int k = 0; k < d; ++k) {
        hipLaunchKernelGGL(chol_column_kernel, dim3((d - k - 1 + 255) / 256 + (k == d - 1 ? 1 : 0)), dim3(256), 0, st, a, d, k, flag);
        if (k + 1 < d)
            hipLaunchKernelGGL(chol_update_kernel, dim3((d - k - 1 + 63) / 64, (d - k - 1 + 3) / 4), dim3(256), 0, st, a, d, k, flag);
    }
    hipLaunchKernelGGL(chol_diag_kernel, dim3((d + 255) / 256), dim3(256), 0, st, a, d);
    failed = 0;
    GDT_CHECK_HIP(hipMemcpyAsync(&failed, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    GDT_CHECK_HIP(hipStreamSynchronize(st));
    return GDT_OK;
}

Operand mat(const double* m, int ld) { Operand o = {}; o.kind = SRC_MAT; o.mat = m; o.ld = ld; return o; }

void gemm(hipStream_t st, const Operand& a, const Operand& b, double* c, int M, int N, int K, double scale) {
    hipLaunchKernelGGL(gemm_kk_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, a, b, c, M, N, K, scale);
}

}  // namespace

extern "C" {

int gdt_whiten_learn_workspace_bytes(int n_vec, int d, int n_pairs, size_t* bytes) {
    GDT_REQUIRE(bytes != nullptr, "bytes");
    Layout L;
    int rc = plan(n_vec, d, n_pairs, L);
    if (rc != GDT_OK) return rc;
    *bytes = L.total;
    return GDT_OK;
}

int gdt_whiten_learn(const float* x, const int* qidx, const int* pidx, int n_vec, int d, int n_pairs, double* m_out, double* p_out,
                     double* eig_out, int* info, void* workspace, size_t workspace_bytes, void* stream_) {
    Layout L;
    int rc = plan(n_vec, d, n_pairs, L);
    if (rc != GDT_OK) return rc;
    GDT_REQUIRE(x && qidx && pidx && m_out && p_out && workspace, "whiten_learn: null buffer");
    if (workspace_bytes < L.total) { gdt_set_error("whiten_learn: workspace too small"); return GDT_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream_;
    char* ws = (char*)(((uintptr_t)workspace + ALIGN - 1) / ALIGN * ALIGN);
    double *S = (double*)(ws + L.s), *Lm = (double*)(ws + L.l), *P0 = (double*)(ws + L.p0), *T1 = (double*)(ws + L.t1),
           *T2 = (double*)(ws + L.t2), *A = (double*)(ws + L.a), *V = (double*)(ws + L.v), *cs = (double*)(ws + L.cs),
           *red = (double*)(ws + L.red);
    int* flag = (int*)(ws + L.flag);
    int* order = (int*)(ws + L.order);
    const unsigned dd_blocks = (unsigned)(((size_t)d * d + 255) / 256);

    // m = mean of the query vectors; S = df df^T / n
    hipLaunchKernelGGL(mean_kernel, dim3((d + 255) / 256), dim3(256), 0, st, x, qidx, n_pairs, d, m_out);
    Operand df = {}; df.kind = SRC_DIFF; df.x = x; df.xd = d; df.q = qidx; df.p = pidx;
    gemm(st, df, df, S, d, d, n_pairs, 1.0 / (double)n_pairs);

    // L = cholesky(S + alpha I), alpha = 0, 1e-10, 1e-9, ... until positive definite (whiten.py:55-70)
    double alpha = 0.0;
    int jitter_steps = 0;
    for (;;) {
        hipLaunchKernelGGL(add_diag_copy_kernel, dim3(dd_blocks), dim3(256), 0, st, S, Lm, d, alpha);
        int failed = 0;
        rc = cholesky_inplace(st, Lm, d, flag, failed);
        if (rc != GDT_OK) return rc;
        if (!failed) break;
        alpha = alpha == 0.0 ? 1e-10 : alpha * 10.0;
        ++jitter_steps;
        if (jitter_steps > 40) { gdt_set_error("whiten_learn: matrix is not positive definite"); return GDT_ERR_INVALID; }
    }
    // P0 = L^-1
    for (int k = 0; k < d; ++k) {
        hipLaunchKernelGGL(tri_inverse_partial_kernel, dim3((d + 255) / 256, TRI_PARTS), dim3(256), 0, st, Lm, P0, T1, d, k);      // T1: scratch
        hipLaunchKernelGGL(tri_inverse_finish_kernel, dim3((d + 255) / 256), dim3(256), 0, st, Lm, P0, T1, d, k);
    }

    // C = sum (x - m)(x - m)^T over ALL vectors; D = P0 C P0^T
    Operand ce = {}; ce.kind = SRC_CENT; ce.x = x; ce.xd = d; ce.mean = m_out;
    gemm(st, ce, ce, T1, d, d, n_vec, 1.0);                                                    // T1 = C (symmetric)
    hipLaunchKernelGGL(transpose_kernel, dim3((d + 31) / 32, (d + 31) / 32), dim3(256), 0, st, P0, T2, d);      // T2 = P0^T  ([k][i] = P0[i][k])
    // preferred: one-sided Jacobi on G = P0 Lc (needs C positive definite and both columns of a pair in LDS)
    int sweeps = 0;
    int c_failed = 1;
    bool converged = false;
    if ((size_t)2 * d * sizeof(double) <= 64 * 1024) {
        hipLaunchKernelGGL(add_diag_copy_kernel, dim3(dd_blocks), dim3(256), 0, st, T1, A, d, 0.0);
        rc = cholesky_inplace(st, A, d, flag, c_failed);
        if (rc != GDT_OK) return rc;
    }
    if (!c_failed) {
        hipLaunchKernelGGL(zero_upper_kernel, dim3(dd_blocks), dim3(256), 0, st, A, d);                        // A = Lc
        gemm(st, mat(A, d), mat(T2, d), V, d, d, d, 1.0);                 // V = G^T: V[i][k] = sum_j Lc[j][i] P0[k][j]
        for (; sweeps < 60; ++sweeps) {
            GDT_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
            for (int r = 0; r < d - 1; ++r)
                hipLaunchKernelGGL(hestenes_round_kernel, dim3(d / 2), dim3(256), (size_t)2 * d * sizeof(double), st, V, d, r, flag, 1e-28);
            int changed = 0;
            GDT_CHECK_HIP(hipMemcpyAsync(&changed, flag, sizeof(int), hipMemcpyDeviceToHost, st));
            GDT_CHECK_HIP(hipStreamSynchronize(st));
            if (!changed) { ++sweeps; converged = true; break; }
        }
        hipLaunchKernelGGL(row_norm2_kernel, dim3(d), dim3(256), 0, st, V, d, cs);                              // cs = eigenvalues (unordered)
        hipLaunchKernelGGL(rank_desc_kernel, dim3((d + 255) / 256), dim3(256), 0, st, cs, (size_t)1, d, order, eig_out);
        hipLaunchKernelGGL(unit_columns_kernel, dim3((d + 31) / 32, (d + 31) / 32), dim3(256), 0, st, V, cs, order, T1, d);
    } else {
        // fallback (C singular, or d too large for the LDS form): two-sided cyclic Jacobi on A = P0 C P0^T, V accumulates the rotations
        gemm(st, mat(T2, d), mat(T1, d), A, d, d, d, 1.0);                                         // A = P0 C        (sum_k P0[i][k] C[k][j])
        hipLaunchKernelGGL(transpose_kernel, dim3((d + 31) / 32, (d + 31) / 32), dim3(256), 0, st, A, T1, d);       // T1 = (P0 C)^T
        gemm(st, mat(T1, d), mat(T2, d), A, d, d, d, 1.0);                                         // A = (P0 C) P0^T (sum_k (P0C)[i][k] P0[j][k])
        hipLaunchKernelGGL(symmetrise_kernel, dim3(dd_blocks), dim3(256), 0, st, A, d);
        hipLaunchKernelGGL(identity_kernel, dim3(dd_blocks), dim3(256), 0, st, V, d);
        for (; sweeps < 40; ++sweeps) {
            hipLaunchKernelGGL(offdiag_kernel, dim3(1), dim3(1024), 0, st, A, d, red);
            double h[2];
            GDT_CHECK_HIP(hipMemcpyAsync(h, red, sizeof(h), hipMemcpyDeviceToHost, st));
            GDT_CHECK_HIP(hipStreamSynchronize(st));
            if (!(h[0] > 1e-26 * h[1])) { converged = true; break; }                               // off-diagonal norm below 1e-13 of the diagonal's
            for (int r = 0; r < d - 1; ++r) {
                hipLaunchKernelGGL(jacobi_angles_kernel, dim3((d / 2 + 255) / 256), dim3(256), 0, st, A, d, r, cs);
                hipLaunchKernelGGL(jacobi_rows_kernel, dim3((d + 255) / 256, d / 2), dim3(256), 0, st, A, d, r, cs);
                hipLaunchKernelGGL(jacobi_cols_kernel, dim3(2 * d), dim3(256), (size_t)d * sizeof(double), st, A, V, d, r, cs);
            }
        }
        hipLaunchKernelGGL(rank_desc_kernel, dim3((d + 255) / 256), dim3(256), 0, st, A, (size_t)d + 1, d, order, eig_out);
        hipLaunchKernelGGL(gather_cols_kernel, dim3(dd_blocks), dim3(256), 0, st, V, order, T1, d);            // T1[k][i] = V[k][order[i]]
    }
    // P = eigvec^T P0 (eigenvectors = columns of T1, eigenvalues decreasing)
    gemm(st, mat(T1, d), mat(P0, d), p_out, d, d, d, 1.0);                                                      // P[i][j] = sum_k T1[k][i] P0[k][j]
    GDT_CHECK_HIP(hipGetLastError());
    GDT_CHECK_HIP(hipStreamSynchronize(st));
    if (info) { info[0] = jitter_steps; info[1] = c_failed ? -sweeps : sweeps; }      // negative: the two-sided fallback ran
    if (!converged) {          // the outputs are written (best effort), but the caller must not take them for an eigen-decomposition
        gdt_set_error("whiten_learn: the Jacobi eigensolver stopped at its sweep cap (" + std::to_string(sweeps) + " sweeps) without converging");
        return GDT_ERR_NOT_CONVERGED;
    }
    return GDT_OK;
}

}  // extern "C"
