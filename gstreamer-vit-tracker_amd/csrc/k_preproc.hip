// k_preproc.hip — pixel stages of the hot path for gfx950.
//
//  * nv12_to_rgb8_kernel: the reference's whole-frame converter
//    (/root/reference/src/nv12_convert.rs:46-169) as one HBM-bound pass: 1.5 B read + 3 B written
//    per pixel, 4 pixels per lane, integer arithmetic identical to the reference (same constants,
//    arithmetic >> 8, clamp).
//  * preproc_kernel: what VitTrack::init / update do first (call sites
//    /root/reference/src/tracker_context.rs:88,90,120): crop a window around the last box, resize it
//    bilinearly, normalise, and emit it directly as the bf16 patch matrix the patch-embed GEMM reads
//    (row = token, column = c*p*p + py*p + px). Only the crop window is ever converted from NV12 —
//    the reference converts the whole 1080p frame (src/pipeline.rs:105) although the tracker
//    consumes one window. Every sampled NV12 pixel goes through the reference's integer formulas.
//
// Float op order is fixed (this file is compiled with -ffp-contract=off and correctly rounded
// sqrt/div) and identical to oracle/vt_oracle.c, so the outputs agree bit for bit.
#include <algorithm>

#include "vt_common.hpp"

// /root/reference/src/nv12_convert.rs:24-29 (table entries) and :124-131 (per pixel)
__device__ __forceinline__ void yuv_to_rgb(int y, int u, int v, int& r, int& g, int& b) {
    const int yv = 298 * (y - 16);
    r = yv + 409 * (v - 128) + 128;
    g = yv - 100 * (u - 128) - 208 * (v - 128) + 128;
    b = yv + 516 * (u - 128) + 128;
    // clamp_u8(x >> 8) written as clamp first, shift second (same value: the arithmetic shift is
    // monotonic). The shift-then-clamp form is pattern-matched by hipcc (ROCm 7.2) into
    // v_ashr_pk_u8_i32, whose upper 16 result bits are not zero on MI355X although the
    // compiler ORs the result as if they were — measured: wrong bytes 2/3 of every packed dword.
    r = min(max(r, 0), 0xffff) >> 8;
    g = min(max(g, 0), 0xffff) >> 8;
    b = min(max(b, 0), 0xffff) >> 8;
}

// One lane converts 4 horizontally adjacent pixels (two UV pairs). Packed NV12, stride == width
// (src/nv12_convert.rs:53-54,105-106). g: index of the 4-pixel group in the frame, row-major.
__device__ __forceinline__ void nv12_rgb_group4(const uint8_t* __restrict__ nv12, int w, int h, uint8_t* __restrict__ rgb,
                                                long g, int groups_per_row) {
    const uint8_t* yp = nv12;
    const uint8_t* uvp = nv12 + (size_t)w * h;
    const int row = (int)(g / groups_per_row);
    const int col0 = (int)(g % groups_per_row) * 4;
    const size_t yrow = (size_t)row * w;
    const size_t uvrow = (size_t)(row >> 1) * w;
    if (col0 + 3 < w && ((w & 3) == 0)) {
        // aligned fast path: 4 B of Y, 4 B of UV, 12 B out
        const uint32_t y4 = *reinterpret_cast<const uint32_t*>(yp + yrow + col0);
        const uint32_t uv4 = *reinterpret_cast<const uint32_t*>(uvp + uvrow + col0);
        int r[4], gg[4], b[4];
        yuv_to_rgb(y4 & 255, uv4 & 255, (uv4 >> 8) & 255, r[0], gg[0], b[0]);
        yuv_to_rgb((y4 >> 8) & 255, uv4 & 255, (uv4 >> 8) & 255, r[1], gg[1], b[1]);
        yuv_to_rgb((y4 >> 16) & 255, (uv4 >> 16) & 255, uv4 >> 24, r[2], gg[2], b[2]);
        yuv_to_rgb(y4 >> 24, (uv4 >> 16) & 255, uv4 >> 24, r[3], gg[3], b[3]);
        uint32_t* o = reinterpret_cast<uint32_t*>(rgb + (yrow + col0) * 3);
        o[0] = r[0] | (gg[0] << 8) | (b[0] << 16) | (r[1] << 24);
        o[1] = gg[1] | (b[1] << 8) | (r[2] << 16) | (gg[2] << 24);
        o[2] = b[2] | (r[3] << 8) | (gg[3] << 16) | (b[3] << 24);
    } else {
        for (int c = col0; c < min(col0 + 4, w); ++c) {
            // UV pair of column (c & ~1): src/nv12_convert.rs:111-113 and the odd tail :152
            const int u = uvp[uvrow + (c & ~1)], v = uvp[uvrow + (c & ~1) + 1];
            int r, gg, b;
            yuv_to_rgb(yp[yrow + c], u, v, r, gg, b);
            uint8_t* o = rgb + (yrow + c) * 3;
            o[0] = (uint8_t)r; o[1] = (uint8_t)gg; o[2] = (uint8_t)b;
        }
    }
}

__global__ __launch_bounds__(256) void nv12_to_rgb8_kernel(const uint8_t* __restrict__ nv12, int w,
                                                           int h, uint8_t* __restrict__ rgb) {
    const int groups_per_row = (w + 3) >> 2;
    const long total = (long)groups_per_row * h;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long)gridDim.x * blockDim.x)
        nv12_rgb_group4(nv12, w, h, rgb, g, groups_per_row);
}

// Wide variant for w % 16 == 0 (1080p: 1920, 4K: 3840): one lane converts a 16 x 2 pixel block - the
// reference's own unit of work is the row PAIR that shares one UV row (src/nv12_convert.rs:59-65).
// Per lane: two 16-B Y loads and ONE 16-B UV load (8 U,V pairs serve 32 pixels), six 16-B stores
// (48 B of RGB per row): every access is a full dwordx4 and a wave touches 1 KiB of contiguous Y /
// 3 KiB of contiguous RGB per row, against 4-B loads and 12-B stores in the general kernel (measured
// there: 2.4 TB/s at 1080p, 3.8 TB/s at 4K of the ~6.3 TB/s a copy reaches). Same integer arithmetic.
// g: index of the 16 x 2 block in the frame (bpr = w / 16 blocks per row pair).
__device__ __forceinline__ void nv12_rgb_block16x2(const uint8_t* __restrict__ nv12, int w, int h, uint8_t* __restrict__ rgb,
                                                   long g, int bpr) {
    const uint8_t* yp = nv12;
    const uint8_t* uvp = nv12 + (size_t)w * h;
    const int rp = (int)(g / bpr), col0 = (int)(g % bpr) << 4;
    const uint4 uv = *reinterpret_cast<const uint4*>(uvp + (size_t)rp * w + col0);
    const uint32_t uvw[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
        const int row = 2 * rp + r2;
        if (row >= h) break;                  // odd height: the last pair has one row
        const uint4 y4 = *reinterpret_cast<const uint4*>(yp + (size_t)row * w + col0);
        const uint32_t yw[4] = {y4.x, y4.y, y4.z, y4.w};
        uint32_t o[12];
#pragma unroll
        for (int q = 0; q < 4; ++q) {         // 4 pixels = one Y dword and one UV dword (2 pairs)
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
        dst[0] = s0; dst[1] = s1; dst[2] = s2;
    }
}

__global__ __launch_bounds__(256) void nv12_to_rgb8_wide_kernel(const uint8_t* __restrict__ nv12, int w,
                                                                int h, uint8_t* __restrict__ rgb) {
    const int bpr = w >> 4;                       // 16-pixel blocks per row
    const long total = (long)bpr * ((h + 1) >> 1);
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total;
         g += (long)gridDim.x * blockDim.x)
        nv12_rgb_block16x2(nv12, w, h, rgb, g, bpr);
}

// ---- n frames per launch (vt_nv12_to_rgb8_batch_device) ------------------------------------------------------------
// A host that still wants RGB for all its cameras (the reference converts every frame, src/pipeline.rs:105) pays one
// launch ramp per call, not per frame: a single 1080p conversion is a 3.9-us kernel bounded by that ramp (0.29 of the HBM
// roof), a batch is one stream of bytes. The frame table travels in the kernel arguments; in[i] == nullptr = the
// reference's all-zero frame for a short buffer (src/nv12_convert.rs:48-50). Same per-pixel arithmetic as above.
//
// WIDE (w % 16 == 0, 16-B aligned frames): the STORE side decides the rate (tools/nv12_batch_probe.hip, 30 / 60 x 1080p):
//   three 16-B stores per lane and row at a 48-B lane stride - every store instruction touches a third of each line it
//   covers - plain 4.0 TB/s, non-temporal 2.7 (partial lines written through); the same bytes handed over through 3 KB of
//   LDS per wave so that every store instruction writes 1 KB of WHOLE contiguous lines: plain 4.6-5.0 TB/s, non-temporal
//   6.2-6.5 TB/s = 0.77-0.81 of the 8 TB/s roof (an output written once and read by somebody else gains nothing from the
//   L2, and whole lines need no read-modify-write on the way out). A wave's 64 blocks of 16 x 2 pixels: converted in
//   registers, written to the wave's LDS at lane * 48 (conflict-free: 12-dword stride), read back as chunk k * 64 + lane
//   and stored at THAT chunk's address (source lane = chunk / 3, piece = chunk % 3) - a wave may straddle a row pair
//   or a frame, every chunk carries its own destination.
template <bool WIDE>
__global__ __launch_bounds__(256) void nv12_to_rgb8_batch_kernel(Nv12Batch bt, int n, int w, int h) {
    if constexpr (!WIDE) {
        const int per_row = (w + 3) >> 2;
        const long per_frame = (long)per_row * h, total = per_frame * n;
        for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
            const int f = (int)(g / per_frame);
            const long gl = g - (long)f * per_frame;
            uint8_t* dst = bt.out[f];
            if (bt.in[f] == nullptr) {                  // short buffer: zero frame
                const int row = (int)(gl / per_row), col0 = (int)(gl % per_row) * 4;
                for (int c = col0; c < min(col0 + 4, w); ++c) {
                    uint8_t* o = dst + ((size_t)row * w + c) * 3;
                    o[0] = 0; o[1] = 0; o[2] = 0;
                }
                continue;
            }
            nv12_rgb_group4(bt.in[f], w, h, dst, gl, per_row);
        }
    } else {
        __shared__ __attribute__((aligned(16))) char lds[4][3072];
        const int bpr = w >> 4, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const long per_frame = (long)bpr * ((h + 1) >> 1), total = per_frame * n;
        char* my = lds[wave];
        for (long g0 = (long)blockIdx.x * blockDim.x + (threadIdx.x & ~63); g0 < total; g0 += (long)gridDim.x * blockDim.x) {
            // this lane's block (clamped past the end: converted, never stored)
            const long gc = g0 + lane < total ? g0 + lane : total - 1;
            const int f = (int)(gc / per_frame);
            const long gl = gc - (long)f * per_frame;
            const uint8_t* src = bt.in[f];
            const int rp = (int)(gl / bpr), col0 = (int)(gl % bpr) << 4;
            u32x4_t uv = {0u, 0u, 0u, 0u};
            if (src) uv = *reinterpret_cast<const u32x4_t*>(src + (size_t)w * h + (size_t)rp * w + col0);
            // destinations of the three chunks this lane stores (row 0 of the chunk's row pair; + w * 3 for row 1)
            uint8_t* dst[3];
            bool live[3], two_rows[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int c = k * 64 + lane, L = c / 3, j = c - 3 * L;
                const long gs = g0 + L;
                live[k] = gs < total;
                const long gsc = live[k] ? gs : total - 1;
                const int fs = (int)(gsc / per_frame);
                const long gls = gsc - (long)fs * per_frame;
                const int rps = (int)(gls / bpr), cs = (int)(gls % bpr) << 4;
                dst[k] = bt.out[fs] + ((size_t)(2 * rps) * w + cs) * 3 + j * 16;
                two_rows[k] = 2 * rps + 1 < h;           // odd height: the last pair has one row
            }
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                const int row = min(2 * rp + r2, h - 1);
                uint32_t o[12];
                if (src) {
                    const u32x4_t y4 = *reinterpret_cast<const u32x4_t*>(src + (size_t)row * w + col0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {         // 4 pixels = one Y dword and one UV dword (2 pairs)
                        int r[4], gg[4], b[4];
                        const int u0 = uv[q] & 255, v0 = (uv[q] >> 8) & 255, u1 = (uv[q] >> 16) & 255, v1 = uv[q] >> 24;
                        yuv_to_rgb(y4[q] & 255, u0, v0, r[0], gg[0], b[0]);
                        yuv_to_rgb((y4[q] >> 8) & 255, u0, v0, r[1], gg[1], b[1]);
                        yuv_to_rgb((y4[q] >> 16) & 255, u1, v1, r[2], gg[2], b[2]);
                        yuv_to_rgb(y4[q] >> 24, u1, v1, r[3], gg[3], b[3]);
                        o[3 * q + 0] = r[0] | (gg[0] << 8) | (b[0] << 16) | (r[1] << 24);
                        o[3 * q + 1] = gg[1] | (b[1] << 8) | (r[2] << 16) | (gg[2] << 24);
                        o[3 * q + 2] = b[2] | (r[3] << 8) | (gg[3] << 16) | (b[3] << 24);
                    }
                } else {                                  // short buffer: zero frame
#pragma unroll
                    for (int e = 0; e < 12; ++e) o[e] = 0u;
                }
                u32x4_t* wl = reinterpret_cast<u32x4_t*>(my + lane * 48);
                wl[0] = u32x4_t{o[0], o[1], o[2], o[3]};
                wl[1] = u32x4_t{o[4], o[5], o[6], o[7]};
                wl[2] = u32x4_t{o[8], o[9], o[10], o[11]};
                __builtin_amdgcn_wave_barrier();          // wave-private LDS: program order + the compiler's lgkmcnt waits
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(my + (k * 64 + lane) * 16);
                    if (live[k] && (r2 == 0 || two_rows[k]))
                        __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(dst[k] + (size_t)r2 * w * 3));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

hipError_t launch_nv12_to_rgb8_batch(const uint8_t* const* d_in, uint8_t* const* d_out, int n, int w, int h, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += VT_NV12_BATCH_MAX) {
        const int m = std::min(VT_NV12_BATCH_MAX, n - i0);
        Nv12Batch bt{};
        bool wide = (w % 16) == 0 && (((size_t)w * h) % 16) == 0;
        for (int i = 0; i < m; ++i) {
            bt.in[i] = d_in[i0 + i]; bt.out[i] = d_out[i0 + i];
            wide = wide && (reinterpret_cast<uintptr_t>(bt.in[i]) % 16) == 0 && (reinterpret_cast<uintptr_t>(bt.out[i]) % 16) == 0;
        }
        const long per_frame = wide ? (long)(w >> 4) * ((h + 1) >> 1) : (long)((w + 3) >> 2) * h;
        const long total = per_frame * m;
        int blocks = (int)std::min<long>((total + 255) / 256, 256 * 16);      // 16 blocks per CU, grid-stride the rest
        if (blocks < 1) blocks = 1;
        if (wide) vt_launch(nv12_to_rgb8_batch_kernel<true>, dim3(blocks), dim3(256), 0, st, bt, m, w, h);
        else vt_launch(nv12_to_rgb8_batch_kernel<false>, dim3(blocks), dim3(256), 0, st, bt, m, w, h);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_nv12_to_rgb8(const uint8_t* nv12, int w, int h, uint8_t* rgb, hipStream_t st) {
    // 16-B accesses need 16-B aligned planes: the UV plane starts at w*h, rows are w bytes
    // below ~16 Mpx the conversion is a 4-10 us kernel bounded by ramp-up, and the 4-pixel kernel's 16x
    // more lanes fill the chip sooner (measured: 1080p 3.9 vs 4.7 us, 4K 9.9 vs 10.2 us); the wide
    // kernel is for larger surfaces (8K: 4.95 TB/s)
    const bool wide = (size_t)w * h >= ((size_t)16 << 20) && (w % 16) == 0 && (((size_t)w * h) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(nv12) % 16) == 0 && (reinterpret_cast<uintptr_t>(rgb) % 16) == 0;
    if (wide) {
        const long total = (long)(w >> 4) * ((h + 1) >> 1);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (blocks < 1) blocks = 1;
        vt_launch(nv12_to_rgb8_wide_kernel, dim3(blocks), dim3(256), 0, st, nv12, w, h, rgb);
        return hipGetLastError();
    }
    const long total = (long)((w + 3) >> 2) * h;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;  // 8 blocks per CU, grid-stride the rest
    if (blocks < 1) blocks = 1;
    vt_launch(nv12_to_rgb8_kernel, dim3(blocks), dim3(256), 0, st, nv12, w, h, rgb);
    return hipGetLastError();
}

// frame pixel (px,py) as float RGB; outside the frame -> 0 (zero padding)
__device__ __forceinline__ void fetch_rgb(const FrameDesc& f, int px, int py, float* rgb, int& miss) {
    if (px < 0 || py < 0 || px >= f.w || py >= f.h) {
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
        return;
    }
    int r, g, b;
    const int sx = px - f.x0, sy = py - f.y0;   // position inside the stored window
    // inside the frame but outside what the caller stored (a window narrower than the crop): black,
    // never an out-of-bounds read. The library's own window planner always covers the crop.
    if ((unsigned)sx >= (unsigned)f.ww || (unsigned)sy >= (unsigned)f.wh) {
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
        miss = 1;
        return;
    }
    if (f.fmt == VT_PIX_RGB8) {
        const uint8_t* p = f.p0 + (size_t)sy * f.s0 + (size_t)sx * 3;
        r = p[0]; g = p[1]; b = p[2];
    } else if (f.fmt == VT_PIX_NV12) {
        const int y = f.p0[(size_t)sy * f.s0 + sx];
        const uint8_t* uv = f.p1 + (size_t)(sy >> 1) * f.s1 + (sx & ~1);   // x0, y0 even
        yuv_to_rgb(y, uv[0], uv[1], r, g, b);
    } else {  // YUY2: Y0 U Y1 V per pixel pair
        const uint8_t* p = f.p0 + (size_t)sy * f.s0 + (size_t)(sx & ~1) * 2;
        yuv_to_rgb(p[(sx & 1) * 2], p[1], p[3], r, g, b);
    }
    rgb[0] = (float)r; rgb[1] = (float)g; rgb[2] = (float)b;
}

// grid: (ceil(size*size/256), nb); one lane per output pixel, 3 channels each.
__global__ __launch_bounds__(256) void preproc_kernel(const FrameDesc* __restrict__ frames,
                                                      StreamState* __restrict__ states,
                                                      bf16_t* __restrict__ patches, int b0,
                                                      int size, int patch, int kpad, int ntok,
                                                      int row_off, float factor, float na0,
                                                      float na1, float na2, float nb0, float nb1,
                                                      float nb2, int is_template) {
    const int b = b0 + blockIdx.y;
    const FrameDesc f = frames[b];
    StreamState& s = states[b];
    // crop geometry — same operations, same order as vto_crop_geometry (oracle/vt_oracle.c)
    const float bx = s.box[0], by = s.box[1], bw = s.box[2], bh = s.box[3];
    const float area = bw * bh;
    const float side = factor * sqrtf(area);
    const float scale = side / (float)size;
    const float cx = bx + 0.5f * bw;
    const float cy = by + 0.5f * bh;
    const float half = 0.5f * side;
    const float x0m = (cx - half) - 0.5f;
    const float y0m = (cy - half) - 0.5f;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix == 0 && !is_template) {
        s.geo[0] = x0m; s.geo[1] = y0m; s.geo[2] = scale; s.geo[3] = side;
        s.frame_w = f.w; s.frame_h = f.h;
    }
    if (pix >= size * size) return;
    const int oy = pix / size, ox = pix % size;
    const float fy = ((float)oy + 0.5f) * scale + y0m;
    const float fx = ((float)ox + 0.5f) * scale + x0m;
    const float fy0 = floorf(fy), fx0 = floorf(fx);
    const float wy = fy - fy0, wx = fx - fx0;
    const int iy = (int)fy0, ix = (int)fx0;
    float p00[3], p01[3], p10[3], p11[3];
    int miss = 0;
    fetch_rgb(f, ix, iy, p00, miss);
    fetch_rgb(f, ix + 1, iy, p01, miss);
    fetch_rgb(f, ix, iy + 1, p10, miss);
    fetch_rgb(f, ix + 1, iy + 1, p11, miss);
    if (miss && !is_template) s.window_miss = s.frames_done + 1;   // every writer stores the same value
    const int grid = size / patch;
    const int token = (oy / patch) * grid + (ox / patch);
    const int kin = (oy % patch) * patch + (ox % patch);
    bf16_t* row = patches + ((size_t)b * ntok + row_off + token) * kpad;
    const float na[3] = {na0, na1, na2}, nb[3] = {nb0, nb1, nb2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = p00[c] + wx * (p01[c] - p00[c]);
        const float bot = p10[c] + wx * (p11[c] - p10[c]);
        const float v = top + wy * (bot - top);
        const float o = v * na[c] + nb[c];
        row[c * patch * patch + kin] = f32_to_bf16(o);
    }
}

// Wide-store variant: one lane produces PX horizontally adjacent output pixels of one patch row for
// all three channels and writes each channel's run with ONE store (PX = 8: 16 B, patch % 8 == 0;
// PX = 2: 4 B, even patch sizes such as 14). The patch matrix is the dominant traffic of this stage
// (0.88 MB out per stream against a ~0.1 MB source window that stays in L1/L2): with one lane per pixel
// the stores were 2-B scatters, 48 per 32-B segment. Same per-pixel arithmetic and tap order as
// preproc_kernel (bit-exact with oracle/vt_oracle.c); neighbouring pixels re-fetch shared taps from
// L1. grid: (ceil(size*size/PX/256), nb).
template <int PX>
__global__ __launch_bounds__(256) void preproc_wide_kernel(const FrameDesc* __restrict__ frames,
                                                           StreamState* __restrict__ states,
                                                           bf16_t* __restrict__ patches, int b0,
                                                           int size, int patch, int kpad, int ntok,
                                                           int row_off, float factor, float na0,
                                                           float na1, float na2, float nb0, float nb1,
                                                           float nb2, int is_template) {
    const int b = b0 + blockIdx.y;
    const FrameDesc f = frames[b];
    StreamState& s = states[b];
    // crop geometry — same operations, same order as vto_crop_geometry (oracle/vt_oracle.c)
    const float bx = s.box[0], by = s.box[1], bw = s.box[2], bh = s.box[3];
    const float area = bw * bh;
    const float side = factor * sqrtf(area);
    const float scale = side / (float)size;
    const float cx = bx + 0.5f * bw;
    const float cy = by + 0.5f * bh;
    const float half = 0.5f * side;
    const float x0m = (cx - half) - 0.5f;
    const float y0m = (cy - half) - 0.5f;
    const int grp = blockIdx.x * blockDim.x + threadIdx.x;        // group of PX pixels
    if (grp == 0 && !is_template) {
        s.geo[0] = x0m; s.geo[1] = y0m; s.geo[2] = scale; s.geo[3] = side;
        s.frame_w = f.w; s.frame_h = f.h;
    }
    const int gpr = size / PX;                                     // groups per output row
    if (grp >= gpr * size) return;
    const int oy = grp / gpr, ox0 = (grp % gpr) * PX;
    const float fy = ((float)oy + 0.5f) * scale + y0m;
    const float fy0 = floorf(fy);
    const float wy = fy - fy0;
    const int iy = (int)fy0;
    const float na[3] = {na0, na1, na2}, nb[3] = {nb0, nb1, nb2};
    bf16_t o[3][PX];
    int miss = 0;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const float fx = ((float)(ox0 + k) + 0.5f) * scale + x0m;
        const float fx0 = floorf(fx);
        const float wx = fx - fx0;
        const int ix = (int)fx0;
        float p00[3], p01[3], p10[3], p11[3];
        fetch_rgb(f, ix, iy, p00, miss);
        fetch_rgb(f, ix + 1, iy, p01, miss);
        fetch_rgb(f, ix, iy + 1, p10, miss);
        fetch_rgb(f, ix + 1, iy + 1, p11, miss);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = p00[c] + wx * (p01[c] - p00[c]);
            const float bot = p10[c] + wx * (p11[c] - p10[c]);
            const float v = top + wy * (bot - top);
            o[c][k] = f32_to_bf16(v * na[c] + nb[c]);
        }
    }
    if (miss && !is_template) s.window_miss = s.frames_done + 1;   // every writer stores the same value
    const int grid = size / patch;
    const int token = (oy / patch) * grid + (ox0 / patch);         // PX divides patch: one token per group
    const int kin = (oy % patch) * patch + (ox0 % patch);
    bf16_t* row = patches + ((size_t)b * ntok + row_off + token) * kpad;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        bf16_t* dst = row + c * patch * patch + kin;
        if constexpr (PX == 8) {
            uint4 v;
            v.x = o[c][0] | ((uint32_t)o[c][1] << 16); v.y = o[c][2] | ((uint32_t)o[c][3] << 16);
            v.z = o[c][4] | ((uint32_t)o[c][5] << 16); v.w = o[c][6] | ((uint32_t)o[c][7] << 16);
            *reinterpret_cast<uint4*>(dst) = v;
        } else {
            *reinterpret_cast<uint32_t*>(dst) = o[c][0] | ((uint32_t)o[c][1] << 16);
        }
    }
}

// Tile variant (size % 64 == 0, patch % 8 == 0): a block produces a 64 x 32 tile of output pixels.
// At the usual scales (0.67 source pixels per output pixel for a 64-px target, 1.6 at 4K) the four
// bilinear taps of neighbouring outputs hit the same few source pixels: the wide kernel fetched and
// converted every source pixel ~8 times (three 1-byte loads + the integer YUV->RGB each time - the
// stage was bound by those scattered byte loads, 46 us for 30 streams). Here the tile's source
// rectangle is fetched ONCE through the same fetch_rgb (formats, frame border, window extent all as
// before), kept in LDS as r | g << 8 | b << 16 | miss << 24, and the lanes interpolate from LDS. A
// tap that lay outside the stored window still raises window_miss only if an output pixel uses it.
// Same arithmetic, same order: bit-exact with the other kernels and the oracle. Rectangles that do
// not fit the 16 KiB buffer take the direct fetches of the wide kernel: that is every tile from a scale of 2
// source pixels per output pixel, i.e. targets from ~130 px at search 384 and ~90 px at search 256 - ordinary
// sizes at 1080p, and a cliff: 30 streams, 1080p, by target size (profiles/r05_preproc_by_target.txt): search 384:
// 64 px 18.6 us, 128 px 25.2, 160 px 92.6, 256 px 83.6; search 256: 64 px 13.5, 96 px and larger 39-40 (2-4 % of a
// pass instead of 0.4 %). Round 5, first attempt: such rectangles staged in 2 or 4 horizontal strips of the tile
// (bit-exact; cfg3 160 px 35 us, cfg2 96 px 23 us) - but every form of the strip loop tried (inside the body, as an
// outer loop over opaque passes, with the 8-pixel run in halves) left hipcc at 76-99 VGPRs and 20 spilled SGPRs
// where this kernel has 62: six or five blocks per CU instead of eight, + 3.8 us on the 64-px benchmark target; not
// shipped. What shipped: the kernel in three buffer tiers (below), one captured graph each, chosen per pass by the
// engine from the boxes the host already knows: 160 px at cfg3 29.5 us, 96 px at cfg2 19.5 us, the 64-px case on
// the unchanged tier-0 kernel.
#define PRE_TILE_W 64
#define PRE_TILE_H 32
// 16 KiB of LDS per block: eight 256-thread blocks per CU (the thread limit), so the 2,160 blocks of a
// 30-stream pass are resident at once - a block is one dependent chain (descriptor -> state -> fetch -> LDS ->
// interpolate -> store, ~7 us) and with 32 KiB (5 blocks per CU) the pass took two rounds of it. A 64-px target
// at 1080p needs ~1,000 source pixels per tile, a 128-px one ~3,900.
#define PRE_TILE_LDS 4096       // source pixels (16 KiB): tier 0, the benchmark's 64-px targets
// Round 5 - larger buffers for larger targets, chosen PER LAUNCH by the engine from the boxes the host already knows
// (vt_engine.hip: one captured graph per tier): tier 1 = 8,192 pixels (32 KiB, five blocks per CU: scales up to ~1.9
// source pixels per output pixel, targets up to ~185 px at search 384 / ~120 px at search 256), tier 2 = 16,384 pixels
// (64 KiB, two blocks per CU: scales up to ~2.75, ~260 / ~175 px). A tile that still does not fit takes the per-pixel
// path below - correct at any size, a 3-4x cliff in time (profiles/r05_preproc_by_target.txt) that tier 0 alone hit
// from ~130-px targets. The tier changes which path a tile takes, never a value: every path is bit-exact.
template <int LDSPX>
__global__ __launch_bounds__(256) void preproc_tile_kernel(const FrameDesc* __restrict__ frames,
                                                           StreamState* __restrict__ states,
                                                           bf16_t* __restrict__ patches, int b0,
                                                           int size, int patch, int kpad, int ntok,
                                                           int row_off, float factor, float na0,
                                                           float na1, float na2, float nb0, float nb1,
                                                           float nb2, int is_template) {
    __shared__ uint32_t src[LDSPX];
    constexpr int PX = 8;
    const int b = b0 + blockIdx.y;
    const FrameDesc f = frames[b];
    StreamState& s = states[b];
    // crop geometry — same operations, same order as vto_crop_geometry (oracle/vt_oracle.c)
    const float bx = s.box[0], by = s.box[1], bw = s.box[2], bh = s.box[3];
    const float area = bw * bh;
    const float side = factor * sqrtf(area);
    const float scale = side / (float)size;
    const float cx = bx + 0.5f * bw;
    const float cy = by + 0.5f * bh;
    const float half = 0.5f * side;
    const float x0m = (cx - half) - 0.5f;
    const float y0m = (cy - half) - 0.5f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && !is_template) {
        s.geo[0] = x0m; s.geo[1] = y0m; s.geo[2] = scale; s.geo[3] = side;
        s.frame_w = f.w; s.frame_h = f.h;
    }
    const int tiles_x = size / PRE_TILE_W;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    // lane -> 8-pixel run of the tile. patch 16 (round 4): the tile is 4 x 2 tokens and a half-wave takes ONE
    // token - lane pair (2 py, 2 py + 1) of 16 patch rows - so that a channel's store instruction writes the
    // 512 contiguous bytes of that token's channel block instead of 16-B pieces of four different patch rows
    // (rows of the patch matrix lie kpad * 2 bytes apart); other patch sizes keep the row-major assignment.
    int oy, ox0;
    if (patch == 16) {
        const int tok = threadIdx.x >> 5, py = (threadIdx.x >> 1) & 15, hx = threadIdx.x & 1;
        oy = ty * PRE_TILE_H + (tok >> 2) * 16 + py;
        ox0 = tx * PRE_TILE_W + (tok & 3) * 16 + hx * PX;
    } else {
        oy = ty * PRE_TILE_H + (threadIdx.x >> 3);
        ox0 = tx * PRE_TILE_W + (threadIdx.x & 7) * PX;
    }
    // source rectangle of the tile: taps of its first and last output pixel (fx, fy grow with ox, oy)
    const int sx_lo = (int)floorf(((float)(tx * PRE_TILE_W) + 0.5f) * scale + x0m);
    const int sx_hi = (int)floorf(((float)(tx * PRE_TILE_W + PRE_TILE_W - 1) + 0.5f) * scale + x0m) + 1;
    const int sy_lo = (int)floorf(((float)(ty * PRE_TILE_H) + 0.5f) * scale + y0m);
    const int sy_hi = (int)floorf(((float)(ty * PRE_TILE_H + PRE_TILE_H - 1) + 0.5f) * scale + y0m) + 1;
    const long sw = (long)sx_hi - sx_lo + 1, sh = (long)sy_hi - sy_lo + 1;
    const bool staged = sw > 0 && sh > 0 && sw * sh <= LDSPX;      // block-uniform
    if (staged) {
        const int n = (int)(sw * sh), w_ = (int)sw;
        // NV12 planes whose rows start on 8-byte boundaries (the library's packed windows: pack_window; whole
        // frames with such strides): the rectangle is fetched in groups of 8 pixels - ONE 8-byte load of Y
        // and ONE of the interleaved UV row (4 pairs) per group, where the per-pixel path issues 24 byte
        // loads - and converted with the same integer formulas. A group that is not entirely inside the
        // frame and the stored window goes through fetch_rgb pixel by pixel (frame border: black; outside the
        // window: black + miss), so every entry of the LDS image is what the per-pixel loop writes.
        const bool fast = f.fmt == VT_PIX_NV12 && (((uintptr_t)f.p0 | (uintptr_t)f.p1 | (uintptr_t)f.s0 | (uintptr_t)f.s1) & 7) == 0;
        if (fast) {
            const int g_lo = (sx_lo - f.x0) >> 3, g_hi = (sx_hi - f.x0) >> 3;     // arithmetic shift: floor for negatives
            const int gpr = g_hi - g_lo + 1, ng = gpr * (int)sh;
            for (int i = threadIdx.x; i < ng; i += 256) {
                const int ry = i / gpr, wx0 = (g_lo + i % gpr) << 3;                // window column of the group
                const int py = sy_lo + ry, wy_ = py - f.y0, px0 = wx0 + f.x0;
                const bool inside = (unsigned)wy_ < (unsigned)f.wh && (unsigned)py < (unsigned)f.h && wx0 >= 0 &&
                                    wx0 + 7 < f.ww && px0 >= 0 && px0 + 7 < f.w;
                uint32_t* dst = src + ry * w_ + (px0 - sx_lo);
                if (inside) {
                    const uint2 y8 = *reinterpret_cast<const uint2*>(f.p0 + (size_t)wy_ * f.s0 + wx0);
                    const uint2 uv8 = *reinterpret_cast<const uint2*>(f.p1 + (size_t)(wy_ >> 1) * f.s1 + wx0);
                    const uint32_t yw[2] = {y8.x, y8.y}, uw[2] = {uv8.x, uv8.y};
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int col = px0 - sx_lo + k;
                        if (col < 0 || col >= w_) continue;
                        const uint32_t pair = uw[k >> 2] >> (((k >> 1) & 1) * 16);   // U, V of the pixel pair
                        int r, g, b;
                        yuv_to_rgb((int)((yw[k >> 2] >> ((k & 3) * 8)) & 255u), (int)(pair & 255u), (int)((pair >> 8) & 255u), r, g, b);
                        dst[k] = (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
                    }
                } else {
                    for (int k = 0; k < 8; ++k) {
                        const int col = px0 - sx_lo + k;
                        if (col < 0 || col >= w_) continue;
                        float p[3];
                        int miss = 0;
                        fetch_rgb(f, px0 + k, py, p, miss);
                        dst[k] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)miss << 24);
                    }
                }
            }
        } else {
            for (int i = threadIdx.x; i < n; i += 256) {
                float p[3];
                int miss = 0;
                fetch_rgb(f, sx_lo + i % w_, sy_lo + i / w_, p, miss);
                src[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)miss << 24);
            }
        }
        __syncthreads();
    }
    const float fy = ((float)oy + 0.5f) * scale + y0m;
    const float fy0 = floorf(fy);
    const float wy = fy - fy0;
    const int iy = (int)fy0;
    const float na[3] = {na0, na1, na2}, nb[3] = {nb0, nb1, nb2};
    const int grid = size / patch;
    const int token = (oy / patch) * grid + (ox0 / patch);         // PX divides patch: one token per group
    const int kin = (oy % patch) * patch + (ox0 % patch);
    bf16_t* row = patches + ((size_t)b * ntok + row_off + token) * kpad;
    int miss = 0;
    if (!staged) {
        // rectangles over 4,096 source pixels (targets from ~130 px at search 384, ~90 px at search 256: see the
        // kernel's header): direct fetches, pixel by pixel, 2-byte stores. Kept out of
        // the staged path's code: inlined into its unrolled loop the 32 fetch_rgb bodies cost 70 VGPRs and
        // with them a block per CU.
#pragma unroll 1
        for (int k = 0; k < PX; ++k) {
            const float fx = ((float)(ox0 + k) + 0.5f) * scale + x0m;
            const float fx0 = floorf(fx);
            const float wx = fx - fx0;
            const int ix = (int)fx0;
            float p00[3], p01[3], p10[3], p11[3];
            fetch_rgb(f, ix, iy, p00, miss);
            fetch_rgb(f, ix + 1, iy, p01, miss);
            fetch_rgb(f, ix, iy + 1, p10, miss);
            fetch_rgb(f, ix + 1, iy + 1, p11, miss);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float top = p00[c] + wx * (p01[c] - p00[c]);
                const float bot = p10[c] + wx * (p11[c] - p10[c]);
                const float v = top + wy * (bot - top);
                row[c * patch * patch + kin + k] = f32_to_bf16(v * na[c] + nb[c]);
            }
        }
        if (miss && !is_template) s.window_miss = s.frames_done + 1;
        return;
    }
    bf16_t o[3][PX];
    const int w_ = (int)sw;
    const uint32_t* r0base = src + (iy - sy_lo) * w_ - sx_lo;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const float fx = ((float)(ox0 + k) + 0.5f) * scale + x0m;
        const float fx0 = floorf(fx);
        const float wx = fx - fx0;
        const uint32_t* r0 = r0base + (int)fx0;
        const uint32_t t00 = r0[0], t01 = r0[1], t10 = r0[w_], t11 = r0[w_ + 1];
        miss |= (int)((t00 | t01 | t10 | t11) >> 24);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p00 = (float)((t00 >> (8 * c)) & 255u), p01 = (float)((t01 >> (8 * c)) & 255u);
            const float p10 = (float)((t10 >> (8 * c)) & 255u), p11 = (float)((t11 >> (8 * c)) & 255u);
            const float top = p00 + wx * (p01 - p00);
            const float bot = p10 + wx * (p11 - p10);
            const float v = top + wy * (bot - top);
            o[c][k] = f32_to_bf16(v * na[c] + nb[c]);
        }
    }
    if (miss && !is_template) s.window_miss = s.frames_done + 1;   // every writer stores the same value
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint4 v;
        v.x = o[c][0] | ((uint32_t)o[c][1] << 16); v.y = o[c][2] | ((uint32_t)o[c][3] << 16);
        v.z = o[c][4] | ((uint32_t)o[c][5] << 16); v.w = o[c][6] | ((uint32_t)o[c][7] << 16);
        *reinterpret_cast<uint4*>(row + c * patch * patch + kin) = v;
    }
}

// The smallest buffer tier (0, 1, 2; 3 = none: per-pixel path) whose tile buffer holds the source rectangle of a 64 x
// 32 output tile of a w x h box, with a few per cent of headroom for a box that grows between the host's knowledge of
// it and the launch. Only a choice of speed.
int preproc_tier_for_box(const ModelDims& d, float w, float h, bool is_template) {
    const float size = (float)(is_template ? d.T : d.S), factor = is_template ? 2.0f : 4.0f;
    const float scale = 1.06f * factor * sqrtf(fmaxf(w * h, 1.0f)) / size;
    const float px = (PRE_TILE_W * scale + 3.0f) * (PRE_TILE_H * scale + 3.0f);
    return px <= PRE_TILE_LDS ? 0 : (px <= 2 * PRE_TILE_LDS ? 1 : (px <= 4 * PRE_TILE_LDS ? 2 : 3));
}

hipError_t launch_preproc(const FrameDesc* frames, StreamState* states, bf16_t* patches,
                          const ModelDims& d, int b0, int nb, bool is_template, hipStream_t st, int tier) {
    const int size = is_template ? d.T : d.S;
    const int row_off = is_template ? 0 : d.nt;
    const float factor = is_template ? 2.0f : 4.0f;
#define PRE_ARGS frames, states, patches, b0, size, d.patch, d.kpad, d.ntok, row_off, factor, d.norm_a[0], \
                 d.norm_a[1], d.norm_a[2], d.norm_b[0], d.norm_b[1], d.norm_b[2], is_template ? 1 : 0
    // store alignment: a run starts at element c*p*p + py*p + px0 of a row of kpad elements
    if (d.patch % 8 == 0 && d.kpad % 8 == 0 && size % PRE_TILE_W == 0 && size % PRE_TILE_H == 0) {
        dim3 grid((size / PRE_TILE_W) * (size / PRE_TILE_H), nb);   // 64 x 32 output tiles, source staged in LDS
        if (tier <= 0) vt_launch(preproc_tile_kernel<PRE_TILE_LDS>, grid, dim3(256), 0, st, PRE_ARGS);
        else if (tier == 1) vt_launch(preproc_tile_kernel<2 * PRE_TILE_LDS>, grid, dim3(256), 0, st, PRE_ARGS);
        else vt_launch(preproc_tile_kernel<4 * PRE_TILE_LDS>, grid, dim3(256), 0, st, PRE_ARGS);
    } else if (d.patch % 8 == 0 && d.kpad % 8 == 0) {          // 16-B stores
        dim3 grid((size * size / 8 + 255) / 256, nb);
        vt_launch(preproc_wide_kernel<8>, grid, dim3(256), 0, st, PRE_ARGS);
    } else if (d.patch % 2 == 0 && d.kpad % 2 == 0) {          // 4-B stores (patch 14)
        dim3 grid((size * size / 2 + 255) / 256, nb);
        vt_launch(preproc_wide_kernel<2>, grid, dim3(256), 0, st, PRE_ARGS);
    } else {
        dim3 grid((size * size + 255) / 256, nb);
        vt_launch(preproc_kernel, grid, dim3(256), 0, st, PRE_ARGS);
    }
#undef PRE_ARGS
    return hipGetLastError();
}
