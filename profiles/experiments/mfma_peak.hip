// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate on MI355X with nothing else in the loop (random operands, 8 independent
// accumulators per wave).  Gives the practical MFMA ceiling (clock under load) the conv kernels are measured against.
// build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NTH>
__global__ __launch_bounds__(NTH) void k(const _Float16* src, float* out, int iters) {
    f16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = *(const f16x8*)(src + (threadIdx.x * 4 + i) * 8);
    for (int i = 0; i < 2; ++i) b[i] = *(const f16x8*)(src + 65536 + (threadIdx.x * 2 + i) * 8);
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[i & 1], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    if (s == 1234.5f) out[0] = s;
}

int main() {
    _Float16* src; float* out;
    std::vector<_Float16> h(1 << 18);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((x >> 8) & 0xffff) / 65536.f - 0.5f); }
    hipMalloc((void**)&src, h.size() * 2); hipMalloc((void**)&out, 64);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves : {4, 8, 16}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (waves == 4) hipLaunchKernelGGL((k<8, 256>), dim3(256), dim3(256), 0, 0, src, out, iters);
            if (waves == 8) hipLaunchKernelGGL((k<8, 512>), dim3(256), dim3(512), 0, 0, src, out, iters);
            if (waves == 16) hipLaunchKernelGGL((k<8, 512>), dim3(512), dim3(512), 0, 0, src, out, iters / 2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 256.0 * (waves == 16 ? 8 : waves) * iters * 8 * 2.0 * 32 * 32 * 16;
            printf("waves/CU %2d: %.3f ms  %.1f TFLOP/s  (implied clock at 1024 flop/clk/SIMD: %.2f GHz)\n", waves, ms, flops / ms / 1e9,
                   flops / (ms * 1e-3) / (256.0 * 4 * 1024) / 1e9);
        }
    }
    return 0;
}
