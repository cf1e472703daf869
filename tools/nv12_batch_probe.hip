// nv12_batch_probe.hip — store-path variants of the n-frames-per-launch NV12 -> RGB8 converter (k_preproc.hip), timed on
// device-resident random frames:   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/nv12_batch_probe.hip -o /tmp/nvb && /tmp/nvb
//   A  16 x 2 pixels per lane, three 16-B stores per row at a 48-B lane stride (every store instruction touches a third
//      of each line it covers), non-temporal
//   B  the same with plain stores
//   C  a wave's 3 KB of a row go through 3 KB of LDS: every store instruction writes 1 KB of whole contiguous lines, plain
//   D  C with non-temporal stores   <- what k_preproc.hip ships (nv12_to_rgb8_batch_kernel<true>; results: profiles/r06_nv12_batch_probe.txt)
// Every variant is checked against variant B byte for byte before it is timed.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gstreamer-vit-tracker_amd/csrc/k_preproc.hip"

thread_local LaunchProbe* vt_launch_probe = nullptr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// the per-lane form of k_preproc.hip's nv12_rgb_block16x2 with the store flavour as a parameter
template <bool NT>
__device__ __forceinline__ void probe_block16x2(const uint8_t* __restrict__ nv12, int w, int h, uint8_t* __restrict__ rgb, long g, int bpr) {
    const uint8_t* yp = nv12;
    const uint8_t* uvp = nv12 + (size_t)w * h;
    const int rp = (int)(g / bpr), col0 = (int)(g % bpr) << 4;
    const uint4 uv = *reinterpret_cast<const uint4*>(uvp + (size_t)rp * w + col0);
    const uint32_t uvw[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
        const int row = 2 * rp + r2;
        if (row >= h) break;
        const uint4 y4 = *reinterpret_cast<const uint4*>(yp + (size_t)row * w + col0);
        const uint32_t yw[4] = {y4.x, y4.y, y4.z, y4.w};
        uint32_t o[12];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int r[4], gg[4], b[4];
            const int u0 = uvw[q] & 255, v0 = (uvw[q] >> 8) & 255, u1 = (uvw[q] >> 16) & 255, v1 = uvw[q] >> 24;
            yuv_to_rgb(yw[q] & 255, u0, v0, r[0], gg[0], b[0]);
            yuv_to_rgb((yw[q] >> 8) & 255, u0, v0, r[1], gg[1], b[1]);
            yuv_to_rgb((yw[q] >> 16) & 255, u1, v1, r[2], gg[2], b[2]);
            yuv_to_rgb(yw[q] >> 24, u1, v1, r[3], gg[3], b[3]);
            o[3 * q + 0] = r[0] | (gg[0] << 8) | (b[0] << 16) | (r[1] << 24);
            o[3 * q + 1] = gg[1] | (b[1] << 8) | (r[2] << 16) | (gg[2] << 24);
            o[3 * q + 2] = b[2] | (r[3] << 8) | (gg[3] << 16) | (b[3] << 24);
        }
        u32x4_t* dst = reinterpret_cast<u32x4_t*>(rgb + ((size_t)row * w + col0) * 3);
        const u32x4_t s0 = {o[0], o[1], o[2], o[3]}, s1 = {o[4], o[5], o[6], o[7]}, s2 = {o[8], o[9], o[10], o[11]};
        if (NT) { __builtin_nontemporal_store(s0, dst); __builtin_nontemporal_store(s1, dst + 1); __builtin_nontemporal_store(s2, dst + 2); }
        else { dst[0] = s0; dst[1] = s1; dst[2] = s2; }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void probe_direct(Nv12Batch bt, int n, int w, int h) {
    const int bpr = w >> 4;
    const long per_frame = (long)bpr * ((h + 1) >> 1), total = per_frame * n;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const int f = (int)(g / per_frame);
        probe_block16x2<NT>(bt.in[f], w, h, bt.out[f], g - (long)f * per_frame, bpr);
    }
}

// a wave's 64 blocks of 16 pixels: converted into registers, written to the wave's 3 KB of LDS at lane * 48, read back
// as chunk k * 64 + lane (16 B) and stored at the address of that chunk: source lane L = chunk / 3, piece j = chunk % 3
template <bool NT>
__global__ __launch_bounds__(256) void probe_lds(Nv12Batch bt, int n, int w, int h) {
    __shared__ __attribute__((aligned(16))) char lds[4][3072];
    const int bpr = w >> 4, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long per_frame = (long)bpr * ((h + 1) >> 1), total = per_frame * n;
    char* my = lds[wave];
    for (long g0 = ((long)blockIdx.x * blockDim.x + (threadIdx.x & ~63)); g0 < total; g0 += (long)gridDim.x * blockDim.x) {
        const long g = g0 + lane;
        const bool live = g < total;
        const long gc = live ? g : total - 1;
        const int f = (int)(gc / per_frame);
        const long gl = gc - (long)f * per_frame;
        const uint8_t* src = bt.in[f];
        const int rp = (int)(gl / bpr), col0 = (int)(gl % bpr) << 4;
        const uint4 uv = *reinterpret_cast<const uint4*>(src + (size_t)w * h + (size_t)rp * w + col0);
        const uint32_t uvw[4] = {uv.x, uv.y, uv.z, uv.w};
        // destinations of this lane's three transposed chunks (the same for both rows but for the row offset)
        uint8_t* dst[3];
        bool dlive[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = k * 64 + lane, L = c / 3, j = c - 3 * L;
            const long gs = g0 + L;
            dlive[k] = gs < total;
            const long gsc = dlive[k] ? gs : total - 1;
            const int fs = (int)(gsc / per_frame);
            const long gls = gsc - (long)fs * per_frame;
            const int rps = (int)(gls / bpr), cs = (int)(gls % bpr) << 4;
            dst[k] = bt.out[fs] + ((size_t)(2 * rps) * w + cs) * 3 + j * 16;
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int row = 2 * rp + r2;
            const bool rlive = row < h;
            const uint4 y4 = *reinterpret_cast<const uint4*>(src + (size_t)(rlive ? row : row - 1) * w + col0);
            const uint32_t yw[4] = {y4.x, y4.y, y4.z, y4.w};
            uint32_t o[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int r[4], gg[4], b[4];
                const int u0 = uvw[q] & 255, v0 = (uvw[q] >> 8) & 255, u1 = (uvw[q] >> 16) & 255, v1 = uvw[q] >> 24;
                yuv_to_rgb(yw[q] & 255, u0, v0, r[0], gg[0], b[0]);
                yuv_to_rgb((yw[q] >> 8) & 255, u0, v0, r[1], gg[1], b[1]);
                yuv_to_rgb((yw[q] >> 16) & 255, u1, v1, r[2], gg[2], b[2]);
                yuv_to_rgb(yw[q] >> 24, u1, v1, r[3], gg[3], b[3]);
                o[3 * q + 0] = r[0] | (gg[0] << 8) | (b[0] << 16) | (r[1] << 24);
                o[3 * q + 1] = gg[1] | (b[1] << 8) | (r[2] << 16) | (gg[2] << 24);
                o[3 * q + 2] = b[2] | (r[3] << 8) | (gg[3] << 16) | (b[3] << 24);
            }
            u32x4_t* wl = reinterpret_cast<u32x4_t*>(my + lane * 48);
            wl[0] = u32x4_t{o[0], o[1], o[2], o[3]};
            wl[1] = u32x4_t{o[4], o[5], o[6], o[7]};
            wl[2] = u32x4_t{o[8], o[9], o[10], o[11]};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(my + (k * 64 + lane) * 16);
                // the odd last row of a frame: row pairs of OTHER lanes may have a second row where this lane's has none;
                // liveness of the destination's row = 2 * rps + r2 < h, the same test per chunk
                u32x4_t* d = reinterpret_cast<u32x4_t*>(dst[k] + (size_t)r2 * w * 3);
                bool ok = dlive[k];
                if (r2 == 1 && (h & 1)) {
                    const int c = k * 64 + lane, L = c / 3;
                    const long gs = g0 + L;
                    const long gls = (gs < total ? gs : total - 1) % per_frame;
                    ok = ok && 2 * (int)(gls / bpr) + 1 < h;
                }
                if (ok) { if (NT) __builtin_nontemporal_store(v, d); else *d = v; }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

int main() {
    const int w = 1920, h = 1080;
    const size_t in_b = ((size_t)w * h * 3 / 2 + 255) & ~(size_t)255, out_b = (size_t)w * h * 3;
    for (int n : {30, 60, 120}) {
        uint8_t *din, *dout, *dref;
        CK(hipMalloc(&din, in_b * n)); CK(hipMalloc(&dout, out_b * n)); CK(hipMalloc(&dref, out_b * n));
        std::vector<uint8_t> host(in_b * n);
        uint32_t seed = 2463534242u;
        for (auto& v : host) { seed ^= seed << 13; seed ^= seed >> 17; seed ^= seed << 5; v = (uint8_t)seed; }
        CK(hipMemcpy(din, host.data(), host.size(), hipMemcpyHostToDevice));
        std::vector<uint8_t> ref(out_b * n), got(out_b * n);
        for (int var = 0; var < 4; ++var) {
            for (int blocks : {2048, 4096, 16384}) {
                int done = 0;
                float best = 1e30f;
                for (int i0 = 0; i0 < n; i0 += VT_NV12_BATCH_MAX) ++done;
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                auto run = [&](uint8_t* outbase) {
                    for (int i0 = 0; i0 < n; i0 += VT_NV12_BATCH_MAX) {
                        const int m = std::min(VT_NV12_BATCH_MAX, n - i0);
                        Nv12Batch bt{};
                        for (int i = 0; i < m; ++i) { bt.in[i] = din + (size_t)(i0 + i) * in_b; bt.out[i] = outbase + (size_t)(i0 + i) * out_b; }
                        switch (var) {
                            case 0: hipLaunchKernelGGL(probe_direct<true>, dim3(blocks), dim3(256), 0, 0, bt, m, w, h); break;
                            case 1: hipLaunchKernelGGL(probe_direct<false>, dim3(blocks), dim3(256), 0, 0, bt, m, w, h); break;
                            case 2: hipLaunchKernelGGL(probe_lds<false>, dim3(blocks), dim3(256), 0, 0, bt, m, w, h); break;
                            default: hipLaunchKernelGGL(probe_lds<true>, dim3(blocks), dim3(256), 0, 0, bt, m, w, h); break;
                        }
                    }
                };
                CK(hipMemset(var == 1 ? dref : dout, 0, out_b * n));
                run(var == 1 ? dref : dout);
                CK(hipDeviceSynchronize());
                if (var == 1 && blocks == 2048) CK(hipMemcpy(ref.data(), dref, ref.size(), hipMemcpyDeviceToHost));
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    for (int it = 0; it < 5; ++it) run(dout);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms / 5);
                }
                const double by = (double)w * h * 4.5 * n;
                printf("n %3d  variant %c  blocks %5d  launches %d: %8.1f us  %.2f TB/s  %.3f of 8 TB/s\n", n, "ABCD"[var], blocks, done,
                       best * 1e3, by / (best * 1e-3) / 1e12, by / (best * 1e-3) / 8e12);
                fflush(stdout);
            }
        }
        // correctness of every variant against B (run after the reference exists)
        for (int var : {0, 2, 3}) {
            CK(hipMemset(dout, 0x5a, out_b * n));
            for (int i0 = 0; i0 < n; i0 += VT_NV12_BATCH_MAX) {
                const int m = std::min(VT_NV12_BATCH_MAX, n - i0);
                Nv12Batch bt{};
                for (int i = 0; i < m; ++i) { bt.in[i] = din + (size_t)(i0 + i) * in_b; bt.out[i] = dout + (size_t)(i0 + i) * out_b; }
                if (var == 0) hipLaunchKernelGGL(probe_direct<true>, dim3(2048), dim3(256), 0, 0, bt, m, w, h);
                else if (var == 2) hipLaunchKernelGGL(probe_lds<false>, dim3(2048), dim3(256), 0, 0, bt, m, w, h);
                else hipLaunchKernelGGL(probe_lds<true>, dim3(2048), dim3(256), 0, 0, bt, m, w, h);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), dout, got.size(), hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
            printf("n %3d  variant %c vs B: %zu bytes differ\n", n, "ABCD"[var], bad);
        }
        CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(dref));
    }
    return 0;
}
