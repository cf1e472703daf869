// k_overlay.hip — the reference's overlay drawing on the NV12 luma plane, on the GPU.
//
// The reference draws into the mapped frame on the CPU right after tracking
// (/root/reference/src/pipeline.rs:125-174): draw_background_nv12, draw_text_nv12,
// draw_rect_nv12, draw_crosshair_nv12 (src/nv12_convert.rs:172-343) and draw_cursor /
// draw_selection (src/drawing.rs:5-50). With the frame resident in HBM the same commands are
// applied here by one launch: one lane per luma pixel walks the command list IN ORDER and applies
// every command that covers its pixel — identical to applying the commands one after another,
// because each command's effect on a pixel depends only on that pixel. The coverage predicates are
// the closed forms of the reference's loops, including its usize wrap-around quirks (a rectangle
// whose right edge x + w is negative extends to the last column). Bit-exact with
// oracle/vt_oracle.c's line-by-line restatements.
#include "vt_common.hpp"

typedef unsigned long long u64;

__device__ __forceinline__ u64 as_usize(int v) { return (u64)(long long)v; }  // i32 as usize
__device__ __forceinline__ u64 sat_sub(u64 a, u64 b) { return a > b ? a - b : 0; }
__device__ __forceinline__ u64 umin(u64 a, u64 b) { return a < b ? a : b; }

// the 5x7 font of src/nv12_convert.rs:257-298
__constant__ uint8_t kGlyphs[40][8] = {
    {'0', 0x0E, 0x11, 0x13, 0x15, 0x19, 0x11, 0x0E}, {'1', 0x04, 0x0C, 0x04, 0x04, 0x04, 0x04, 0x0E},
    {'2', 0x0E, 0x11, 0x01, 0x06, 0x08, 0x10, 0x1F}, {'3', 0x0E, 0x11, 0x01, 0x06, 0x01, 0x11, 0x0E},
    {'4', 0x02, 0x06, 0x0A, 0x12, 0x1F, 0x02, 0x02}, {'5', 0x1F, 0x10, 0x1E, 0x01, 0x01, 0x11, 0x0E},
    {'6', 0x06, 0x08, 0x10, 0x1E, 0x11, 0x11, 0x0E}, {'7', 0x1F, 0x01, 0x02, 0x04, 0x08, 0x08, 0x08},
    {'8', 0x0E, 0x11, 0x11, 0x0E, 0x11, 0x11, 0x0E}, {'9', 0x0E, 0x11, 0x11, 0x0F, 0x01, 0x02, 0x0C},
    {'.', 0x00, 0x00, 0x00, 0x00, 0x00, 0x0C, 0x0C}, {':', 0x00, 0x0C, 0x0C, 0x00, 0x0C, 0x0C, 0x00},
    {'-', 0x00, 0x00, 0x00, 0x1F, 0x00, 0x00, 0x00}, {' ', 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00},
    {'F', 0x1F, 0x10, 0x1E, 0x10, 0x10, 0x10, 0x10}, {'P', 0x1E, 0x11, 0x1E, 0x10, 0x10, 0x10, 0x10},
    {'S', 0x0E, 0x11, 0x10, 0x0E, 0x01, 0x11, 0x0E}, {'T', 0x1F, 0x04, 0x04, 0x04, 0x04, 0x04, 0x04},
    {'R', 0x1E, 0x11, 0x1E, 0x14, 0x12, 0x11, 0x11}, {'A', 0x0E, 0x11, 0x1F, 0x11, 0x11, 0x11, 0x11},
    {'C', 0x0E, 0x11, 0x10, 0x10, 0x10, 0x11, 0x0E}, {'K', 0x11, 0x12, 0x14, 0x18, 0x14, 0x12, 0x11},
    {'I', 0x0E, 0x04, 0x04, 0x04, 0x04, 0x04, 0x0E}, {'N', 0x11, 0x19, 0x15, 0x13, 0x11, 0x11, 0x11},
    {'G', 0x0E, 0x11, 0x10, 0x17, 0x11, 0x11, 0x0E}, {'E', 0x1F, 0x10, 0x1E, 0x10, 0x10, 0x10, 0x1F},
    {'L', 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x1F}, {'O', 0x0E, 0x11, 0x11, 0x11, 0x11, 0x11, 0x0E},
    {'D', 0x1C, 0x12, 0x11, 0x11, 0x11, 0x12, 0x1C}, {'%', 0x19, 0x1A, 0x04, 0x04, 0x08, 0x0B, 0x13},
    {'s', 0x00, 0x00, 0x0E, 0x10, 0x0E, 0x01, 0x1E}, {'c', 0x00, 0x00, 0x0E, 0x10, 0x10, 0x11, 0x0E},
    {'o', 0x00, 0x00, 0x0E, 0x11, 0x11, 0x11, 0x0E}, {'r', 0x00, 0x00, 0x16, 0x19, 0x10, 0x10, 0x10},
    {'e', 0x00, 0x00, 0x0E, 0x11, 0x1F, 0x10, 0x0E}, {'m', 0x00, 0x00, 0x1A, 0x15, 0x15, 0x11, 0x11},
    {'t', 0x08, 0x08, 0x1C, 0x08, 0x08, 0x09, 0x06}, {'k', 0x10, 0x10, 0x12, 0x14, 0x18, 0x14, 0x12},
    {'n', 0x00, 0x00, 0x16, 0x19, 0x11, 0x11, 0x11}, {'v', 0x00, 0x00, 0x11, 0x11, 0x11, 0x0A, 0x04},
};

__device__ __forceinline__ bool glyph_bit(char ch, int row, int col) {
    for (int g = 0; g < 40; ++g)
        if ((char)kGlyphs[g][0] == ch) return (kGlyphs[g][1 + row] >> (4 - col)) & 1;
    return false;  // unknown characters are skipped (src/nv12_convert.rs:302), the cursor still advances
}

// does command c touch luma pixel (px, py)? (px < W, py < H)
__device__ __forceinline__ bool covers(const vt_draw_cmd& c, u64 px, u64 py, u64 W, u64 H) {
    switch (c.type) {
        case VT_DRAW_BACKGROUND: {  // src/nv12_convert.rs:325-343
            const u64 x = as_usize(c.x), y = as_usize(c.y);
            return px >= x && px < umin(x + as_usize(c.w), W) && py >= y && py < umin(y + as_usize(c.h), H);
        }
        case VT_DRAW_TEXT: {        // src/nv12_convert.rs:246-322
            const u64 x = as_usize(c.x), y = as_usize(c.y), sc = as_usize(c.p);
            if (sc == 0 || px < x || py < y) return false;
            const u64 dx = px - x, dy = py - y, row = dy / sc;
            if (row >= 7) return false;
            const u64 ci = dx / (6 * sc), col = (dx % (6 * sc)) / sc;
            if (col >= 5 || ci >= sizeof(c.text)) return false;
            for (u64 k = 0; k <= ci; ++k)            // the string ends at the first NUL
                if (c.text[k] == 0) return false;
            return glyph_bit(c.text[ci], (int)row, (int)col);
        }
        case VT_DRAW_RECT: {        // src/nv12_convert.rs:172-213
            const u64 x1 = (u64)(c.x > 0 ? c.x : 0), y1 = (u64)(c.y > 0 ? c.y : 0);
            const u64 x2 = umin(as_usize(c.x + c.w), sat_sub(W, 1));
            const u64 y2 = umin(as_usize(c.y + c.h), sat_sub(H, 1));
            const u64 th = as_usize(c.p);
            const bool in_x = px >= x1 && px <= x2, in_y = py >= y1 && py <= y2;
            if (in_x && ((py >= y1 && py - y1 < th) || (py <= y2 && y2 - py < th))) return true;
            if (in_y && ((px >= x1 && px - x1 < th) || (px <= x2 && x2 - px < th))) return true;
            return false;
        }
        case VT_DRAW_CROSSHAIR: {   // src/nv12_convert.rs:216-243
            const u64 cx = (u64)(c.x > 0 ? c.x : 0), cy = (u64)(c.y > 0 ? c.y : 0), s = as_usize(c.p);
            if (py == cy && px >= sat_sub(cx, s) && px <= umin(cx + s, W - 1)) return true;
            if (px == cx && py >= sat_sub(cy, s) && py <= umin(cy + s, H - 1)) return true;
            return false;
        }
        case VT_DRAW_CURSOR: {      // src/drawing.rs:5-23
            const long long xc = c.x < 0 ? 0 : (c.x > (long long)W - 1 ? (long long)W - 1 : c.x);
            const long long yc = c.y < 0 ? 0 : (c.y > (long long)H - 1 ? (long long)H - 1 : c.y);
            const u64 x = (u64)xc, y = (u64)yc;
            if (py == y && px >= sat_sub(x, 25) && px <= umin(x + 25, W - 1) &&
                !(px >= sat_sub(x, 5) && px <= x + 5)) return true;
            if (px == x && py >= sat_sub(y, 25) && py <= umin(y + 25, H - 1) &&
                !(py >= sat_sub(y, 5) && py <= y + 5)) return true;
            return false;
        }
        case VT_DRAW_SELECTION: {   // src/drawing.rs:25-50: start = (x, y), cursor = (w, h)
            const int mnx = c.x < c.w ? c.x : c.w, mny = c.y < c.h ? c.y : c.h;
            const int mxx = c.x > c.w ? c.x : c.w, mxy = c.y > c.h ? c.y : c.h;
            const u64 x1 = (u64)(mnx > 0 ? mnx : 0), y1 = (u64)(mny > 0 ? mny : 0);
            const u64 x2 = umin(as_usize(mxx), W - 1), y2 = umin(as_usize(mxy), H - 1);
            if ((py == y1 || py == y2) && px >= x1 && px <= x2 && (px / 6) % 2 == 0) return true;
            if ((px == x1 || px == x2) && py >= y1 && py <= y2 && (py / 6) % 2 == 0) return true;
            return false;
        }
        default: return false;
    }
}

__global__ __launch_bounds__(256) void overlay_kernel(uint8_t* __restrict__ yplane, int width,
                                                      int height, int stride,
                                                      const vt_draw_cmd* __restrict__ cmds, int n) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= width || py >= height) return;
    uint8_t v = 0;
    bool loaded = false;
    for (int i = 0; i < n; ++i) {
        const vt_draw_cmd& c = cmds[i];
        if (!covers(c, (u64)px, (u64)py, (u64)width, (u64)height)) continue;
        if (!loaded) { v = yplane[(size_t)py * stride + px]; loaded = true; }
        if (c.type == VT_DRAW_BACKGROUND)
            v = (uint8_t)(((unsigned)v * (unsigned)(255 - (c.value & 255))) / 255u);
        else
            v = (c.type == VT_DRAW_CURSOR || c.type == VT_DRAW_SELECTION) ? 255 : (uint8_t)c.value;
    }
    if (loaded) yplane[(size_t)py * stride + px] = v;
}

hipError_t launch_overlay(uint8_t* yplane, int width, int height, int stride, const vt_draw_cmd* d_cmds,
                          int n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    dim3 grid((width + 63) / 64, (height + 3) / 4);
    vt_launch(overlay_kernel, grid, dim3(256), 0, st, yplane, width, height, stride, d_cmds, n);
    return hipGetLastError();
}

// ---- packed RGB8 surface (src/drawing_rgb.rs:4-129) -----------------------------------------------------
// Same scheme; every write goes through set_pixel_rgb_color's bounds test, so coverage is the plain
// geometric shape clipped to the frame. value = 0xRRGGBB (text: r = g = b = value & 255).
__device__ __forceinline__ bool covers_rgb(const vt_draw_cmd& c, int px, int py, int W, int H) {
    switch (c.type) {
        case VT_DRAW_BACKGROUND: {  // :30-53 (fill with 30)
            const u64 xs = (u64)(c.x > 0 ? c.x : 0), xe = umin(as_usize(c.x + c.w), (u64)W);
            const u64 ys = (u64)(c.y > 0 ? c.y : 0), ye = umin(as_usize(c.y + c.h), (u64)H);
            return xe >= xs && (u64)px >= xs && (u64)px < xe && (u64)py >= ys && (u64)py < ye;
        }
        case VT_DRAW_TEXT: {        // :86-104
            const int sc = c.p;
            if (sc <= 0) return false;
            const long long dx = (long long)px - c.x, dy = (long long)py - c.y;
            if (dx < 0 || dy < 0) return false;
            const long long row = dy / sc, ci = dx / (6LL * sc), col = (dx % (6LL * sc)) / sc;
            if (row >= 7 || col >= 5 || ci >= (long long)sizeof(c.text)) return false;
            for (long long k = 0; k <= ci; ++k)
                if (c.text[k] == 0) return false;
            return glyph_bit(c.text[ci], (int)row, (int)col);
        }
        case VT_DRAW_RECT: {        // :55-66
            const long long rx = (long long)px - c.x, ry = (long long)py - c.y, th = c.p;
            const bool in_w = rx >= 0 && rx < c.w, in_h = ry >= 0 && ry < c.h;
            // rows y+t and y+rh-1-t for i in 0..rw; columns x+t and x+rw-1-t for i in 0..rh
            if (in_w && ((ry >= 0 && ry < th) || (c.h - 1 - ry >= 0 && c.h - 1 - ry < th))) return true;
            if (in_h && ((rx >= 0 && rx < th) || (c.w - 1 - rx >= 0 && c.w - 1 - rx < th))) return true;
            return false;
        }
        case VT_DRAW_CROSSHAIR: {   // :68-73
            const long long dx = (long long)px - c.x, dy = (long long)py - c.y;
            return (dy == 0 && dx >= -(long long)c.p && dx <= c.p) || (dx == 0 && dy >= -(long long)c.p && dy <= c.p);
        }
        case VT_DRAW_CURSOR: {      // :75-84, arms 5..=25
            const long long dx = (long long)px - c.x, dy = (long long)py - c.y;
            const long long ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
            return (dy == 0 && ax >= 5 && ax <= 25) || (dx == 0 && ay >= 5 && ay <= 25);
        }
        case VT_DRAW_SELECTION: {   // :106-129: start = (x, y), cursor = (w, h)
            int x1 = c.x < c.w ? c.x : c.w, y1 = c.y < c.h ? c.y : c.h;
            int x2 = c.x > c.w ? c.x : c.w, y2 = c.y > c.h ? c.y : c.h;
            x1 = x1 > 0 ? x1 : 0; y1 = y1 > 0 ? y1 : 0;
            x2 = x2 < W - 1 ? x2 : W - 1; y2 = y2 < H - 1 ? y2 : H - 1;
            if ((py == y1 || py == y2) && px >= x1 && px <= x2 && (px / 6) % 2 == 0) return true;
            if ((px == x1 || px == x2) && py >= y1 && py <= y2 && (py / 6) % 2 == 0) return true;
            return false;
        }
        default: return false;
    }
}

__global__ __launch_bounds__(256) void overlay_rgb_kernel(uint8_t* __restrict__ rgb, int width, int height,
                                                          int stride, const vt_draw_cmd* __restrict__ cmds,
                                                          int n) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= width || py >= height) return;
    int r = 0, g = 0, b = 0;
    bool hit = false;
    for (int i = 0; i < n; ++i) {
        const vt_draw_cmd& c = cmds[i];
        if (!covers_rgb(c, px, py, width, height)) continue;
        hit = true;
        switch (c.type) {
            case VT_DRAW_BACKGROUND: r = g = b = 30; break;
            case VT_DRAW_TEXT: r = g = b = c.value & 255; break;
            case VT_DRAW_CURSOR: r = 0; g = 255; b = 0; break;
            case VT_DRAW_SELECTION: r = 255; g = 255; b = 0; break;
            default: r = (c.value >> 16) & 255; g = (c.value >> 8) & 255; b = c.value & 255; break;
        }
    }
    if (hit) {
        uint8_t* p = rgb + (size_t)py * stride + (size_t)px * 3;
        p[0] = (uint8_t)r; p[1] = (uint8_t)g; p[2] = (uint8_t)b;
    }
}

hipError_t launch_overlay_rgb(uint8_t* rgb, int width, int height, int stride, const vt_draw_cmd* d_cmds, int n,
                              hipStream_t st) {
    if (n <= 0) return hipSuccess;
    dim3 grid((width + 63) / 64, (height + 3) / 4);
    vt_launch(overlay_rgb_kernel, grid, dim3(256), 0, st, rgb, width, height, stride, d_cmds, n);
    return hipGetLastError();
}
