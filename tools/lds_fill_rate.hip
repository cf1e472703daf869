// lds_fill_rate.hip - how fast can ONE CU pull L2-resident operand tiles into LDS with LDS-DMA (global_load_lds_dwordx4) when
// all 256 CUs do it at once? The number that decides whether a GEMM tile smaller than 256 x 256 (more operand bytes per flop) can
// keep the matrix pipe busy. 512 threads per workgroup (one per CU: 128 KiB of LDS), every wave issues 1-KiB
// pieces (64 lanes x 16 B, whole 128-B lines of 8 rows) round-robin into a ring, DEPTH pieces per wave in flight (counted
// vmcnt); the source is a region all workgroups share (L2 hits after the first touch).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_fill_rate.hip -o /tmp/lds_fill_rate && /tmp/lds_fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int DEPTH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void fill(const char* src, size_t region, int iters, unsigned long long* clk, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a wave's pieces: 1 KiB each, DEPTH slots of the ring per wave
    char* ring = smem + wave * DEPTH * 1024;
    size_t off = ((size_t)blockIdx.x * 4096 + (size_t)wave * 65536 + (size_t)lane * 16) % region;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                             (__attribute__((address_space(3))) void*)(ring + d * 1024), 16, 0, 0);
            off += 1024 * WAVES;
            off = off >= region ? off - region : off;
            wait_vm<DEPTH - 1>();
        }
    }
    wait_vm<0>();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
    if (smem[threadIdx.x] == 123 && iters < 0) sink[0] = 1.0f;
}

template <int DEPTH, int WAVES>
static void run(const char* src, size_t region, int grid) {
    const int iters = 2000;
    unsigned long long* clk; float* sink;
    (void)hipMalloc(&clk, grid * 8); (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fill<DEPTH, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((fill<DEPTH, WAVES>), dim3(grid), dim3(WAVES * 64), 128 * 1024, 0, src, region, iters, clk, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), clk, grid * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= grid;
    const double bytes = (double)iters * DEPTH * 1024 * WAVES;       // per workgroup
    printf("region %6.1f MiB, %d workgroups x %d waves, %d pieces per wave in flight: %7.1f GB/s per CU, %6.2f TB/s chip, %5.1f B per s_memtime tick (100 MHz x ... see ms: %.3f)\n",
           region / 1048576.0, grid, WAVES, DEPTH, bytes / (ms * 1e-3) * 1e-9, bytes * grid / (ms * 1e-3) * 1e-12, bytes / c, ms);
    (void)hipFree(clk); (void)hipFree(sink);
}

int main() {
    const size_t cap = 64u << 20;
    char* src; (void)hipMalloc(&src, cap); (void)hipMemset(src, 1, cap);
    for (size_t region : {(size_t)1 << 20, (size_t)4 << 20, (size_t)32 << 20}) {
        run<2, 8>(src, region, 256);
        run<4, 8>(src, region, 256);
        run<8, 8>(src, region, 256);
        run<16, 8>(src, region, 256);
        run<8, 4>(src, region, 256);
        run<16, 4>(src, region, 256);
    }
    run<8, 8>(src, (size_t)1 << 20, 32);       // an eighth of the chip
    return 0;
}
