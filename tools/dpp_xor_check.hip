// dpp_xor_check.hip - the register-only 32-lane xor butterfly of ln_row (vt_common.hpp: v_permlane16_swap + DPP) against the
// __shfl_xor (ds_bpermute) butterfly it replaced: same partners in the same order (16, 8, 4, 2, 1), so the same bits.
// Build: hipcc --offload-arch=gfx950 -O3 tools/dpp_xor_check.hip -o /tmp/dpp_xor_check && /tmp/dpp_xor_check (on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../gstreamer-vit-tracker_amd/csrc/vt_common.hpp"

__global__ void k(const float* x, float* y, float* z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float v = x[i];
    y[i] = half_wave_sum(v);
    float s = v;
    for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    z[i] = s;
}

int main() {
    const int n = 64 * 4096;
    std::vector<float> hx(n), hy(n), hz(n);
    unsigned seed = 12345u;
    for (auto& v : hx) { seed = seed * 1664525u + 1013904223u; v = ((int)(seed >> 8) % 200001 - 100000) * 1.37e-3f; }
    float *dx, *dy, *dz;
    if (hipMalloc(&dx, n * 4) != hipSuccess || hipMalloc(&dy, n * 4) != hipSuccess || hipMalloc(&dz, n * 4) != hipSuccess) return 2;
    (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, nullptr, dx, dy, dz);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    (void)hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hz.data(), dz, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += memcmp(&hy[i], &hz[i], 4) != 0;
    for (int i = 0; i < 40 && bad; i += 3) printf("lane %d: x %.9g  dpp %.9g  shfl %.9g\n", i, hx[i], hy[i], hz[i]);
    printf("half_wave_sum vs __shfl_xor butterfly: %d of %d lanes differ\n", bad, n);
    return bad ? 1 : 0;
}
