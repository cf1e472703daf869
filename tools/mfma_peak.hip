// What does the matrix pipe of this MI355X sustain? A register-only MFMA loop (no memory traffic), 256 workgroups
// of 512 threads (2 waves per SIMD, like the 256x256 GEMM kernels), v_mfma_f32_16x16x32_bf16 and 32x32x16.
// Prints TFLOP/s over HIP events, and the shader clock: s_memtime ticks over s_memrealtime (100 MHz) ticks.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Measurement tool only - not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND, int INDEP>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* out, unsigned long long* clk) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f / (1 + i + threadIdx.x)); }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (KIND == 0) {
        f32x4 acc[INDEP];
        for (int k = 0; k < INDEP; ++k) acc[k] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < INDEP; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
        for (int k = 0; k < INDEP; ++k) s += acc[k][0] + acc[k][3];
    } else {
        f32x16 acc[INDEP];
        for (int k = 0; k < INDEP; ++k)
            for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int k = 0; k < INDEP; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k], 0, 0, 0);
        for (int k = 0; k < INDEP; ++k) s += acc[k][0] + acc[k][15];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND, int INDEP>
static void run(const char* name, int grid, int iters, double flop_per_mfma, int threads = 512) {
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)grid * 512 * 4); hipMalloc(&clk, (size_t)grid * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {            // third run reported: clocks have settled under the load
        hipEventRecord(e0); 
        mfma_loop<KIND, INDEP><<<grid, threads>>>(iters, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * grid);
    hipMemcpy(h.data(), clk, (size_t)grid * 16, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < grid; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
    const int wps = threads / 256;        // waves per SIMD
    const double flops = (double)grid * (threads / 64) * iters * INDEP * flop_per_mfma;
    const double ticks = c / grid;
    printf("%-32s grid %4d x %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  s_memtime ticks: %.2f per MFMA issued on a SIMD, %.1f MHz tick rate "
           "(ticks / event time), s_memtime/s_memrealtime %.2f\n", name, grid, wps, ms, flops / ms * 1e-9,
           ticks / ((double)iters * INDEP * wps), ticks / (ms * 1e3), c / r);
    hipFree(out); hipFree(clk);
}

int main() {
    const int iters = 20000;
    run<0, 8>("mfma_f32_16x16x32_bf16 x8 indep", 256, iters, 2.0 * 16 * 16 * 32);
    run<1, 4>("mfma_f32_32x32x16_bf16 x4 indep", 256, iters, 2.0 * 32 * 32 * 16);
    run<0, 8>("mfma_f32_16x16x32_bf16 x8 indep", 32, iters, 2.0 * 16 * 16 * 32);      // 1/8 of the chip: power headroom
    run<0, 8>("mfma_f32_16x16x32_bf16 x8 indep", 256, iters * 4, 2.0 * 16 * 16 * 32);  // ~ 4x longer: sustained
    run<0, 8>("mfma_f32_16x16x32_bf16 x8 indep", 32, iters, 2.0 * 16 * 16 * 32, 256);      // one wave per SIMD
    run<0, 8>("mfma_f32_16x16x32_bf16 x8 indep", 256, iters, 2.0 * 16 * 16 * 32, 256);
    run<0, 8>("mfma_f32_16x16x32_bf16 x8 indep", 256, iters, 2.0 * 16 * 16 * 32, 1024);    // four waves per SIMD
    return 0;
}
