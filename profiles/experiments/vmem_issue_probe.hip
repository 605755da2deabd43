// Micro-benchmark: what does ONE vector-memory load cost a wave that otherwise issues MFMAs back to back (one wave per SIMD, in order)?
// A step = 16 independent v_mfma_f32_32x32x16_f16 (512 pipe cycles) + NL global loads of an L2-resident weight stream, consumed two
// steps later (ring of three), exactly the shape of the weight stream of conv3x3_halo_c.hip.  Reported: shader cycles per step
// (s_memtime) for NL = 0, 1, 2, 4, 8, loads clumped / spread between the MFMAs, 64-bit vector addresses / scalar base + 32-bit offset,
// dwordx4 / dwordx2 / dword per lane, and LDS reads of the same size for comparison.
// build: hipcc -O3 --offload-arch=gfx950 vmem_issue_probe.hip -o vmem_issue_probe ; run: ./vmem_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

// KIND 0: global dwordx4, 64-bit vaddr   1: global dwordx4, scalar base + 32-bit lane offset   2: dwordx2   3: dword   4: LDS ds_read_b128
template <int NL, int KIND, bool SPREAD, unsigned WINDOW, bool WAVEPRIV = false, int NDS = 0, int CLUMP_AT = 0>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ w, float* out, unsigned long long* cyc, int steps3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 1024 + 64; i += 256) ((v4i*)smem)[i] = ((const v4i*)w)[i];          // the 16 KB window (+ 1 KB of slack) in LDS for KIND 4
    __syncthreads();
    constexpr int NLR = NL ? NL : 1;
    v4i ring[3][NLR];
    constexpr int NA = NDS ? NDS : 2;
    f16x8 a[NA];
    for (int i = 0; i < NA; ++i) a[i] = *(const f16x8*)(w + i * 4096 + lane * 16);
    const char* base = w + (WAVEPRIV ? wave * (WINDOW / 4) : 0);       // WAVEPRIV: the four waves of a CU walk disjoint quarters (no L1 sharing)
    unsigned lo = lane * 16;
    asm volatile("" : "+v"(lo));
    auto ld = [&](int slot, int k, int step) {
        const unsigned off = ((unsigned)(step * NLR + k) * 1024u) & (KIND == 4 ? 0x3fffu : (WAVEPRIV ? WINDOW / 4 : WINDOW) - 1u);      // uniform; WINDOW bytes walked in order
        if (KIND == 0) ring[slot][k] = *(const v4i*)(base + (size_t)off + (size_t)lo);
        if (KIND == 1) ring[slot][k] = *(const v4i*)(base + off + lo);
        if (KIND == 2) { const v2i t = *(const v2i*)(base + off + (lo >> 1)); ring[slot][k][0] = t[0]; ring[slot][k][1] = t[1]; }
        if (KIND == 3) ring[slot][k][0] = *(const int*)(base + off + (lo >> 2));
        if (KIND == 4) ring[slot][k] = *(const v4i*)(smem + off + lo);
    };
    if (NL)
        for (int k = 0; k < NLR; ++k) { ld(0, k, 0); ld(1, k, 1); ring[2][k] = ring[0][k]; }
    else
        for (int s = 0; s < 3; ++s) ring[s][0] = *(const v4i*)(w + s * 1024 + lane * 16);
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < steps3; ++it) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const f16x8 b = __builtin_bit_cast(f16x8, ring[s][m % NLR]);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[NDS ? m * NDS / 16 : m & 1], b, acc[m], 0, 0, 0);
                // (NDS) the activation fragment is re-loaded from LDS in place behind its last MFMA, as the conv kernel does
                if (NDS && (m + 1) % (16 / NA) == 0) a[m * NDS / 16] = *(const f16x8*)(smem + (((it * 3 + s) * NA + m) & 15) * 1024 + lo);
                if (NL) {
                    // re-load the slot the previous step has finished with: slot (s + 2) % 3, data of step it * 3 + s + 2
                    if (SPREAD) { if (m % (16 / NL) == 0) ld((s + 2) % 3, m / (16 / NL), it * 3 + s + 2); }
                    else if (m == CLUMP_AT) { for (int k = 0; k < NL; ++k) ld((s + 2) % 3, k, it * 3 + s + 2); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sacc = 0.f;
    for (int i = 0; i < 16; ++i) sacc += acc[i][0] + acc[i][9];
    if (sacc == 1234.5f) out[0] = sacc;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NL, int KIND, bool SPREAD, unsigned WINDOW, bool WAVEPRIV = false, int NDS = 0, int CLUMP_AT = 0>
void run(const char* name, const char* w, float* out, unsigned long long* cyc) {
    const int steps3 = 2000, grid = 256;
    hipFuncSetAttribute((const void*)probe<NL, KIND, SPREAD, WINDOW, WAVEPRIV, NDS, CLUMP_AT>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<NL, KIND, SPREAD, WINDOW, WAVEPRIV, NDS, CLUMP_AT>), dim3(grid), dim3(256), 100 * 1024, 0, w, out, cyc, steps3);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    const double per_step = s / h.size() / (steps3 * 3.0);
    printf("%-46s NL %d: %7.1f cycles per step (16 MFMAs = 512)%s", name, NL, per_step, NL ? "" : "\n");
    if (NL) printf("  -> %5.1f per load\n", (per_step - 512.0) / NL);
}

int main() {
    char* w; float* out; unsigned long long* cyc;
    std::vector<_Float16> h(1 << 20);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((x >> 8) & 0xffff) / 65536.f - 0.5f); }
    hipMalloc((void**)&w, h.size() * 2); hipMalloc((void**)&out, 64); hipMalloc((void**)&cyc, 1024 * 8);
    hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice);
#define ROW(KIND, SPREAD, WINDOW, name)                                  \
    run<1, KIND, SPREAD, WINDOW>(name, w, out, cyc);                     \
    run<2, KIND, SPREAD, WINDOW>(name, w, out, cyc);                     \
    run<4, KIND, SPREAD, WINDOW>(name, w, out, cyc);                     \
    run<8, KIND, SPREAD, WINDOW>(name, w, out, cyc);
    run<0, 0, false, 16384u>("no loads", w, out, cyc);
    ROW(1, true, 16384u, "dwordx4, 16 KB window (L1 hits)")
    ROW(1, false, (1u << 21), "dwordx4, 2 MB window (L2 hits), clumped")
    ROW(1, true, (1u << 21), "dwordx4, 2 MB window (L2 hits), spread")
    ROW(0, true, (1u << 21), "dwordx4 vaddr64, 2 MB window, spread")
#define ROWP(KIND, SPREAD, WINDOW, name)                                       \
    run<1, KIND, SPREAD, WINDOW, true>(name, w, out, cyc);                     \
    run<2, KIND, SPREAD, WINDOW, true>(name, w, out, cyc);                     \
    run<4, KIND, SPREAD, WINDOW, true>(name, w, out, cyc);                     \
    run<8, KIND, SPREAD, WINDOW, true>(name, w, out, cyc);
    ROWP(1, true, (1u << 21), "dwordx4, 2 MB, each wave its own quarter, spread")
    ROWP(1, false, (1u << 21), "dwordx4, 2 MB, each wave its own quarter, clumped")
#define ROWD(KIND, SPREAD, WINDOW, name)                                       \
    run<1, KIND, SPREAD, WINDOW, true, 8>(name, w, out, cyc);                  \
    run<2, KIND, SPREAD, WINDOW, true, 8>(name, w, out, cyc);                  \
    run<4, KIND, SPREAD, WINDOW, true, 8>(name, w, out, cyc);                  \
    run<8, KIND, SPREAD, WINDOW, true, 8>(name, w, out, cyc);
    run<0, 1, true, (1u << 21), true, 8>("no global loads, 8 ds_read_b128 fragment re-loads", w, out, cyc);
    ROWD(1, true, (1u << 21), "dwordx4 own quarter + 8 LDS fragment re-loads, spread")
    // prefetch distance: the loads of a step are consumed from the start of the step after next; issued behind MFMA 0 / 8 / 15 of their
    // step they have 2 / 1.5 / 1 steps (1024 / 768 / 512 MFMA cycles) to arrive
    run<2, 1, false, (1u << 21), true, 8, 0>("2 loads + 8 LDS, issued behind MFMA 0 (2 steps ahead)", w, out, cyc);
    run<2, 1, false, (1u << 21), true, 8, 8>("2 loads + 8 LDS, issued behind MFMA 8 (1.5 steps ahead)", w, out, cyc);
    run<2, 1, false, (1u << 21), true, 8, 15>("2 loads + 8 LDS, issued behind MFMA 15 (1 step ahead)", w, out, cyc);
    run<4, 1, false, (1u << 21), true, 8, 0>("4 loads + 8 LDS, issued behind MFMA 0 (2 steps ahead)", w, out, cyc);
    run<4, 1, false, (1u << 21), true, 8, 8>("4 loads + 8 LDS, issued behind MFMA 8 (1.5 steps ahead)", w, out, cyc);
    run<4, 1, false, (1u << 21), true, 8, 15>("4 loads + 8 LDS, issued behind MFMA 15 (1 step ahead)", w, out, cyc);
    ROW(2, true, (1u << 21), "dwordx2, 2 MB window, spread")
    ROW(3, true, (1u << 21), "dword, 2 MB window, spread")
    ROW(4, true, 16384u, "ds_read_b128, spread")
    return 0;
}
