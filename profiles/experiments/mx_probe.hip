// Probe for the block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4) on MI355X: (1) operand lane / bit layout checked with exact data
// against a host evaluation, A = fp4 (e2m1), B = fp6 (e2m3), per-lane E8M0 scales; (2) sustained rates of the instruction mixes the
// compensated-fp16 convolution ("f16c" precision mode) is built on: fp16 32x32x16 alone, MX alone, and 2 fp16 + 1 MX per 32 k-values.
// build: hipcc -O3 --offload-arch=gfx950 mx_probe.hip -o mx_probe ; run: ./mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static float fp4_val(int c) { static const float t[8] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f}; return (c & 8) ? -t[c & 7] : t[c & 7]; }
static float fp6_val(int c) {   // e2m3, bias 1
    const int e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m * 0.125f : std::ldexp(1.f + m / 8.f, e - 1);
    return (c & 32) ? -v : v;
}

__global__ void one_mfma(const v8i* a, const v8i* b, f32x16* c, const int* sa, const int* sb) {
    v8i A = a[threadIdx.x], B = b[threadIdx.x];
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 4, 2, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    c[threadIdx.x] = acc;
}

// MODE 0: fp16 only (NACC x per iteration); 1: MX fp4 x fp6 only; 2: per accumulator 2 fp16 + 1 MX (fp4 x fp6); 3: MX fp8 x fp8 only;
// 4: 2 fp16 + 1 MX fp8 x fp8; 5: MX fp6 x fp6 only
template <int MODE, int NTH>
__global__ __launch_bounds__(NTH) void rate(const int* src, float* out, int iters) {
    constexpr int NACC = 8;
    f16x8 a[4], b[2];
    v8i am[4], bm[2];
    for (int i = 0; i < 4; ++i) { a[i] = *(const f16x8*)(src + (threadIdx.x * 4 + i) * 4); am[i] = *(const v8i*)(src + 32768 + (threadIdx.x * 4 + i) * 8); }
    for (int i = 0; i < 2; ++i) { b[i] = *(const f16x8*)(src + 16384 + (threadIdx.x * 2 + i) * 4); bm[i] = *(const v8i*)(src + 65536 + (threadIdx.x * 2 + i) * 8); }
    const int sa = 127 - (threadIdx.x & 3), sb = 127 - ((threadIdx.x >> 2) & 3);
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0 || MODE == 2 || MODE == 4) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i & 1], acc[i], 0, 0, 0);
                if (MODE != 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + 1) & 3], b[i & 1], acc[i], 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am[i & 3], bm[i & 1], acc[i], 4, 2, 0, sa, 0, sb);
            if (MODE == 3 || MODE == 4) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am[i & 3], bm[i & 1], acc[i], 0, 0, 0, sa, 0, sb);
            if (MODE == 5) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am[i & 3], bm[i & 1], acc[i], 2, 2, 0, sa, 0, sb);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    if (s == 1234.5f) out[0] = s;
}

int main() {
    // ------------------------------------------------------------------ (1) layout
    std::vector<int> Ac(32 * 64), Bc(64 * 32), SA(32 * 2), SB(32 * 2);
    unsigned x = 2024;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (x >> 10); };
    for (auto& v : Ac) v = rnd() & 15;
    for (auto& v : Bc) v = rnd() & 63;
    for (auto& v : SA) v = 120 + (rnd() % 12);
    for (auto& v : SB) v = 121 + (rnd() % 10);
    std::vector<int> ha(64 * 8, 0), hb(64 * 8, 0), hsa(64), hsb(64);
    for (int l = 0; l < 64; ++l) {
        const int r = l & 31, kb = l >> 5;
        unsigned char abytes[32] = {0}, bbytes[32] = {0};
        for (int i = 0; i < 32; ++i) {
            const int k = 32 * kb + i;
            abytes[i >> 1] |= (unsigned char)(Ac[r * 64 + k] << (4 * (i & 1)));
            const int bit = 6 * i, code = Bc[k * 32 + r];
            bbytes[bit >> 3] |= (unsigned char)(code << (bit & 7));
            if ((bit & 7) > 2) bbytes[(bit >> 3) + 1] |= (unsigned char)(code >> (8 - (bit & 7)));
        }
        memcpy(&ha[l * 8], abytes, 32); memcpy(&hb[l * 8], bbytes, 32);
        hsa[l] = SA[r * 2 + kb]; hsb[l] = SB[r * 2 + kb];
    }
    int *da, *db, *dsa, *dsb; float* dc;
    hipMalloc((void**)&da, 64 * 32); hipMalloc((void**)&db, 64 * 32); hipMalloc((void**)&dc, 64 * 64); hipMalloc((void**)&dsa, 256); hipMalloc((void**)&dsb, 256);
    hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (f32x16*)dc, dsa, dsb);
    std::vector<float> hc(64 * 16);
    hipMemcpy(hc.data(), dc, 64 * 64, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; ++l)
        for (int reg = 0; reg < 16; ++reg) {
            const int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
            double ref = 0;
            for (int k = 0; k < 64; ++k)
                ref += (double)fp4_val(Ac[row * 64 + k]) * std::ldexp(1.0, SA[row * 2 + (k >> 5)] - 127) *
                       (double)fp6_val(Bc[k * 32 + col]) * std::ldexp(1.0, SB[col * 2 + (k >> 5)] - 127);
            maxerr = std::fmax(maxerr, std::fabs(ref - hc[l * 16 + reg])); maxref = std::fmax(maxref, std::fabs(ref));
        }
    printf("layout check (A fp4 rows on lanes, k-block = lane>>5, little-endian packing, scale byte 0 per lane): max|err| %.3g of max|ref| %.3g -> %s\n",
           maxerr, maxref, maxerr <= 1e-6 * maxref ? "OK" : "MISMATCH");

    // ------------------------------------------------------------------ (2) rates
    std::vector<int> h(1 << 18);
    for (auto& v : h) v = (int)(rnd() * 2654435761u);
    // keep fp16 operands finite: clear the top exponent bit of every half
    for (int i = 0; i < 32768; ++i) h[i] &= 0xBFFFBFFF;
    // keep fp8 (e4m3) operands away from NaN (0x7f / 0xff): clear bit 6 of every byte in the MX operand area
    for (int i = 32768; i < (1 << 18); ++i) h[i] &= 0xBFBFBFBF;
    int* src; float* out;
    hipMalloc((void**)&src, h.size() * 4); hipMalloc((void**)&out, 64);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 8000;
    const char* names[6] = {"fp16 32x32x16 only", "MX fp4 x fp6 only", "2 fp16 + 1 MX(fp4 x fp6)", "MX fp8 x fp8 only", "2 fp16 + 1 MX(fp8 x fp8)", "MX fp6 x fp6 only"};
    for (int mode = 0; mode < 6; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL((rate<0, 512>), dim3(256), dim3(512), 0, 0, src, out, iters); break;
                case 1: hipLaunchKernelGGL((rate<1, 512>), dim3(256), dim3(512), 0, 0, src, out, iters); break;
                case 2: hipLaunchKernelGGL((rate<2, 512>), dim3(256), dim3(512), 0, 0, src, out, iters); break;
                case 3: hipLaunchKernelGGL((rate<3, 512>), dim3(256), dim3(512), 0, 0, src, out, iters); break;
                case 4: hipLaunchKernelGGL((rate<4, 512>), dim3(256), dim3(512), 0, 0, src, out, iters); break;
                case 5: hipLaunchKernelGGL((rate<5, 512>), dim3(256), dim3(512), 0, 0, src, out, iters); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // "k-values" retired per accumulator per iteration: fp16 only 16; MX only 64; mixes: 32 real k (2 x 16 fp16 + the 64-wide MX carrying both corrections)
            const double groups = 256.0 * 8 * iters * 8;       // (accumulator, iteration) groups
            const double ns_per_group = ms * 1e6 / (iters * 8.0) ;   // per wave: 8 groups per iteration, 2 waves per SIMD
            printf("%-28s %.3f ms   %.1f ns per group per wave-pair-slot (2 waves/SIMD)   fp16-equivalent conv rate: %.1f TFLOP/s\n", names[mode], ms,
                   ns_per_group, (mode == 0 ? 16 : mode == 1 || mode == 3 || mode == 5 ? 64 : 32) * 2.0 * 32 * 32 * groups / ms / 1e9);
        }
    return 0;
}
