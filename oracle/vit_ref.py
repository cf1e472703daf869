"""NumPy restatement of the tracker below the VitTrack::{init, update} boundary.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

What it follows:
  * NV12 -> RGB and crop/resize/normalise/decode: oracle/vt_oracle.c through ctypes (the NV12
    part restates /root/reference/src/nv12_convert.rs:8-169).
  * `init` / `update` call shape: /root/reference/src/tracker_context.rs:88,90,120 (init returns
    nothing the caller uses; update returns {success, score, bbox}).
  * The network itself (patch-embed, joint template+search encoder, centre head) has NO reference
    definition — the reference runs an un-vendored RKNN model. It follows this
    build's specification in DESIGN.md §2/§3. PARITY UNPINNED against the reference.

Quantisation points mirror the HIP path ("numerical specification v3"):
GEMM operands are bf16 (weights are stored bf16; activations are rounded to bf16 exactly where the
HIP kernels round them), accumulation, LayerNorm statistics and softmax are float32. The residual
stream x is kept as a 3-BYTE PAIR, hi = bf16(x) and lo8 = clamp(rint((x - hi) * 2^12), -127, 127) as a signed
byte (x = hi + lo8 * 2^-12: an absolute quantum of 2^-12 beside bf16's relative 2^-9; numerical specification v3,
round 6 - until round 5 the low half was a second bf16), and the two LayerNorms of a block are folded into the GEMM that consumes them: the GEMM's
A operand is hi itself, its weights are W' = bf16(gamma * W), and the epilogue applies
    y[m][n] = rstd[m] * (sum_k hi[m][k] W'[n][k] - mean[m] * s[n]) + c[n],
    s[n] = sum_k W'[n][k],  c[n] = sum_k beta[k] W[n][k] + bias[n]
with mean / rstd taken from the float32 value x had before it was split. What remains different
between the two implementations is float32 summation order (and exp2 / GELU last-bit differences).

`ROUNDING = False` turns every bf16 rounding into the identity (the folded form is then the plain
pre-LN transformer in float32): that mode is what tests/test_oracle_crosscheck.py compares with the
independent torch.nn.functional formulation in oracle/torch_ref.py.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np
from scipy.special import erf

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        c_u8p = ctypes.POINTER(ctypes.c_uint8)
        c_fp = ctypes.POINTER(ctypes.c_float)
        _lib.vto_nv12_to_rgb8.argtypes = [c_u8p, ctypes.c_size_t, ctypes.c_size_t,
                                          ctypes.c_size_t, c_u8p, ctypes.c_int]
        _lib.vto_nv12_to_rgb8.restype = ctypes.c_int
        _lib.vto_nv12_bytes_read.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
        _lib.vto_nv12_bytes_read.restype = ctypes.c_size_t
        _lib.vto_yuv_to_rgb_px.argtypes = [ctypes.c_uint8] * 3 + [c_u8p]
        _lib.vto_crop_geometry.argtypes = [c_fp, ctypes.c_float, ctypes.c_int, c_fp]
        _lib.vto_preproc.argtypes = [c_u8p, c_u8p] + [ctypes.c_int] * 5 + [
            c_fp, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp, c_fp,
            ctypes.POINTER(ctypes.c_uint16)]
        sz, i32, u8 = ctypes.c_size_t, ctypes.c_int32, ctypes.c_uint8
        _lib.vto_draw_rect_nv12.argtypes = [c_u8p, sz, sz, i32, i32, i32, i32, sz, u8]
        _lib.vto_draw_crosshair_nv12.argtypes = [c_u8p, sz, sz, i32, i32, i32, u8]
        _lib.vto_draw_text_nv12.argtypes = [c_u8p, sz, sz, ctypes.c_char_p, sz, sz, sz, u8]
        _lib.vto_draw_background_nv12.argtypes = [c_u8p, sz, sz, sz, sz, sz, sz, u8]
        _lib.vto_draw_cursor.argtypes = [c_u8p, sz, sz, i32, i32]
        _lib.vto_draw_selection.argtypes = [c_u8p, sz, sz, i32, i32, i32, i32]
        _lib.vto_draw_background_rgb.argtypes = [c_u8p, sz, sz, i32, i32, i32, i32]
        _lib.vto_draw_rect_rgb.argtypes = [c_u8p, sz, sz, i32, i32, i32, i32, i32, u8, u8, u8]
        _lib.vto_draw_crosshair_rgb.argtypes = [c_u8p, sz, sz, i32, i32, i32, u8, u8, u8]
        _lib.vto_draw_cursor_rgb.argtypes = [c_u8p, sz, sz, i32, i32]
        _lib.vto_draw_text_rgb.argtypes = [c_u8p, sz, sz, ctypes.c_char_p, i32, i32, i32, u8]
        _lib.vto_draw_selection_rgb.argtypes = [c_u8p, sz, sz, i32, i32, i32, i32]
        _lib.vto_decode.argtypes = [c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_int,
                                    c_fp, ctypes.POINTER(ctypes.c_int32)]
    return _lib


def _u8p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


# ---- bf16 ---------------------------------------------------------------------------------

def f32_to_bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))
            >> np.uint32(16)).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(
        np.float32)


ROUNDING = True     # False: every bf16 rounding of the NETWORK becomes the identity (cross-check mode)


def bf16r(x):
    """round float32 values to the nearest bf16, returned as float32"""
    if not ROUNDING:
        return np.ascontiguousarray(x, dtype=np.float32)
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


# ---- colour conversion (reference stage) -----------------------------------------------------

def nv12_to_rgb8(nv12: np.ndarray, w: int, h: int, nthreads: int = 1):
    """≙ nv12_full_to_rgb_parallel (src/nv12_convert.rs:46). -> ((h,w,3) uint8, status)."""
    nv12 = np.ascontiguousarray(nv12, np.uint8)
    out = np.empty((h, w, 3), np.uint8)
    st = lib().vto_nv12_to_rgb8(_u8p(nv12), nv12.size, w, h, _u8p(out), nthreads)
    return out, st


def yuv_px(y, u, v):
    o = np.zeros(3, np.uint8)
    lib().vto_yuv_to_rgb_px(int(y), int(u), int(v), _u8p(o))
    return tuple(int(x) for x in o)


# ---- overlay drawing (reference stage: src/nv12_convert.rs:172-343, src/drawing.rs:5-50) ----------

def draw(nv12: np.ndarray, w: int, h: int, cmds) -> np.ndarray:
    """apply (kind, x, y, w, h, p, value, text) commands in order with the line-by-line C
    restatements; kinds: 0 background, 1 text, 2 rect, 3 crosshair, 4 cursor, 5 selection"""
    buf = np.ascontiguousarray(nv12, np.uint8).reshape(-1).copy()
    L, p = lib(), _u8p(buf)
    for (kind, x, y, cw, ch, pp, value, text) in cmds:
        if kind == 0:
            L.vto_draw_background_nv12(p, w, h, x, y, cw, ch, value)
        elif kind == 1:
            L.vto_draw_text_nv12(p, w, h, text.encode(), x, y, pp, value)
        elif kind == 2:
            L.vto_draw_rect_nv12(p, w, h, x, y, cw, ch, pp, value)
        elif kind == 3:
            L.vto_draw_crosshair_nv12(p, w, h, x, y, pp, value)
        elif kind == 4:
            L.vto_draw_cursor(p, w, h, x, y)
        elif kind == 5:
            L.vto_draw_selection(p, w, h, x, y, cw, ch)
    return buf


def draw_rgb(img: np.ndarray, cmds) -> np.ndarray:
    """the packed-RGB variants (src/drawing_rgb.rs); value = 0xRRGGBB for rect / crosshair"""
    out = np.ascontiguousarray(img, np.uint8).copy()
    h, w, _ = out.shape
    L, p = lib(), _u8p(out)
    for (kind, x, y, cw, ch, pp, value, text) in cmds:
        r, g, b = (value >> 16) & 255, (value >> 8) & 255, value & 255
        if kind == 0:
            L.vto_draw_background_rgb(p, w, h, x, y, cw, ch)
        elif kind == 1:
            L.vto_draw_text_rgb(p, w, h, text.encode(), x, y, pp, value & 255)
        elif kind == 2:
            L.vto_draw_rect_rgb(p, w, h, x, y, cw, ch, pp, r, g, b)
        elif kind == 3:
            L.vto_draw_crosshair_rgb(p, w, h, x, y, pp, r, g, b)
        elif kind == 4:
            L.vto_draw_cursor_rgb(p, w, h, x, y)
        elif kind == 5:
            L.vto_draw_selection_rgb(p, w, h, x, y, cw, ch)
    return out


# ---- frames ---------------------------------------------------------------------------------

class Frame:
    """RGB8 (H,W,3) view (≙ ArrayView3<u8>, src/pipeline_ir.rs:142) or NV12 planes."""

    def __init__(self, fmt, p0, p1, w, h, s0, s1):
        self.fmt, self.p0, self.p1, self.w, self.h, self.s0, self.s1 = fmt, p0, p1, w, h, s0, s1

    @staticmethod
    def rgb8(arr):
        arr = np.ascontiguousarray(arr, np.uint8)
        h, w, _ = arr.shape
        return Frame(0, arr, None, w, h, w * 3, 0)

    @staticmethod
    def yuy2(arr, w, h):
        """packed 4:2:2 (Y0 U Y1 V), rows of 2*w bytes"""
        arr = np.ascontiguousarray(arr, np.uint8).reshape(-1)
        return Frame(2, arr, None, w, h, 2 * w, 0)

    @staticmethod
    def nv12(buf, w, h):
        """packed NV12 buffer, stride == width (src/nv12_convert.rs:53-54)"""
        buf = np.ascontiguousarray(buf, np.uint8).reshape(-1)
        return Frame(1, buf[: w * h], buf[w * h:], w, h, w, w)


def crop_geometry(box, factor, out_size):
    b = np.asarray(box, np.float32)
    geo = np.zeros(4, np.float32)
    lib().vto_crop_geometry(_fp(b), ctypes.c_float(factor), out_size, _fp(geo))
    return geo


def preproc(frame: Frame, box, factor, out_size, patch, kpad, norm_a, norm_b):
    """-> patch matrix [(out_size/patch)^2, kpad] of bf16 bits"""
    g = out_size // patch
    out = np.zeros((g * g, kpad), np.uint16)
    b = np.asarray(box, np.float32)
    na = np.ascontiguousarray(norm_a, np.float32)
    nb = np.ascontiguousarray(norm_b, np.float32)
    p1 = _u8p(frame.p1) if frame.p1 is not None else None
    lib().vto_preproc(_u8p(frame.p0), p1, frame.w, frame.h, frame.s0, frame.s1, frame.fmt, _fp(b),
                      ctypes.c_float(factor), out_size, patch, kpad, _fp(na), _fp(nb),
                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)))
    return out


# ---- the network ------------------------------------------------------------------------------

def row_stats(x, eps):
    """float32 two-pass mean / rstd of every row -> (mean [M,1], rstd [M,1])"""
    x = x.astype(np.float32)
    mean = x.mean(axis=-1, keepdims=True, dtype=np.float32)
    xc = x - mean
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=np.float32)
    rstd = (np.float32(1.0) / np.sqrt(var + np.float32(eps))).astype(np.float32)
    return mean.astype(np.float32), rstd


def layernorm(x, g, b, eps):
    mean, rstd = row_stats(x, eps)
    return ((x.astype(np.float32) - mean) * rstd) * g.reshape(1, -1) + b.reshape(1, -1)


# numerical specification v3 (round 6): the low half of the residual pair is ONE byte - a signed integer in
# units of 2^-12
LO_SHIFT = 12
LO_Q = np.float32(2.0 ** -LO_SHIFT)


def split_residual(v):
    """the residual stream as the 3-byte pair the HIP path stores: hi = bf16(v) and lo8 = clamp(rint((v - hi) * 2^12),
    -127, 127) as a signed byte; returned as (hi, lo8 * 2^-12) in float32. Every step is exact but the two roundings
    (v - hi is exact: hi is v rounded to 8 significant bits; the scaling is a power of two; rint is round-to-nearest-even;
    hi + lo8 * 2^-12 is exact in float32 for |v| < 2^11), so an implementation cannot differ from this one by anything
    but the value of v it starts from. |v - hi| <= ulp(hi) / 2, i.e. at most 64 quanta for |v| < 8: the stored value is
    within half a quantum of v. For 8 <= |v| < 16 the remainder reaches 128 quanta next to a bf16 tie, where the clamp
    costs at most one quantum; beyond 16 it leaves part of the low half behind (the value degrades towards bf16, it
    never wraps). tests/test_weights_and_oracle_model.py holds known answers for each of these cases."""
    v = v.astype(np.float32)
    hi = bf16r(v)
    if not ROUNDING:
        return hi, np.zeros_like(hi)
    lo8 = np.clip(np.rint((v - hi).astype(np.float32) * np.float32(2.0 ** LO_SHIFT)), -127.0, 127.0).astype(np.float32)
    return hi, (lo8 * LO_Q).astype(np.float32)


def fold_layernorm(w, gamma, beta, bias):
    """LayerNorm folded into the GEMM that consumes it: W' = bf16(gamma * W) (one float32 multiply,
    one rounding), s[n] = sum_k W'[n][k], c[n] = sum_k beta[k] W[n][k] + bias[n] (float32)"""
    wf = bf16r((w * gamma.reshape(1, -1)).astype(np.float32))
    s = wf.sum(axis=1, dtype=np.float32)
    c = (w @ beta.reshape(-1).astype(np.float32) + bias.reshape(-1)).astype(np.float32)
    return wf, s, c


def folded_linear(hi, mean, rstd, fold):
    """rstd * (hi W'^T - mean s) + c, float32"""
    wf, s, c = fold
    return (rstd * ((hi @ wf.T) - mean * s.reshape(1, -1)) + c.reshape(1, -1)).astype(np.float32)


def gelu(x):
    x = x.astype(np.float32)
    return (np.float32(0.5) * x * (np.float32(1.0) + erf(x * np.float32(0.7071067811865476)))
            ).astype(np.float32)


QK_SCALE = np.float32(0.125 * 1.4426950408889634)   # 1/sqrt(64) * log2(e): float32(log2 e) / 8


SOFTMAX_REF = "max"    # diagnostic switch (tools/diag_taps.py): "zero" = no max subtraction


def attention(q, k, v, heads):
    """q,k,v [N, H*64] float32 holding bf16 values; q pre-scaled by QK_SCALE, so q.k is the softmax
    exponent in log2 units -> [N, H*64] f32. The probabilities are rounded to bf16 for the P.V
    product and the row sum is taken over those rounded values (as the HIP kernels do)."""
    out = np.empty_like(q)
    for h in range(heads):
        sl = slice(h * 64, (h + 1) * 64)
        s = q[:, sl] @ k[:, sl].T
        m = s.max(axis=1, keepdims=True) if SOFTMAX_REF == "max" else np.float32(0.0)
        p = bf16r(np.exp2((s - m).astype(np.float32)).astype(np.float32))
        l = p.sum(axis=1, keepdims=True, dtype=np.float32)
        out[:, sl] = (p @ v[:, sl]) / l
    return out


def im2col3x3(t, grid):
    """t [grid*grid, C] -> [grid*grid, 9*C], column (ky*3+kx)*C + c, zero padding"""
    c = t.shape[1]
    g = np.zeros((grid + 2, grid + 2, c), np.float32)
    g[1:-1, 1:-1] = t.reshape(grid, grid, c)
    cols = [g[ky:ky + grid, kx:kx + grid].reshape(grid * grid, c)
            for ky in range(3) for kx in range(3)]
    return np.concatenate(cols, axis=1)


def parse_vtwb(raw: bytes):
    """The oracle's OWN reader of the "VTWB0001" weight blob (layout: DESIGN.md §2; the product's
    writer is gstreamer-vit-tracker_amd/weights.py and its reader csrc/vt_engine.hip `index_blob` —
    neither is imported here, so a layout mistake on either side shows up as a parity failure).
      [0,8) magic; [8,88) 20 x int32: version, patch, template, search, dim, heads, layers, mlp_dim,
      head_ch, kpad, n_tensors, seed; [88,120) 8 x float32: norm_a[3], norm_b[3], success
      threshold, LayerNorm eps; table at 256, 64 B per entry: name[32], dtype u32 (0 f32, 1 bf16),
      rows u32, cols u32, pad u32, offset u64, nbytes u64.
    -> (header dict, {name: float32 or uint16 2-D array})"""
    import struct
    if len(raw) < 256 or raw[:8] != b"VTWB0001":
        raise ValueError("not a VTWB0001 blob")
    iv = struct.unpack_from("<20i", raw, 8)
    fv = struct.unpack_from("<8f", raw, 88)
    keys = ("version", "patch", "template", "search", "dim", "heads", "layers", "mlp_dim",
            "head_ch", "kpad", "n_tensors", "seed")
    hdr = dict(zip(keys, iv))
    if hdr["version"] != 1:
        raise ValueError(f"VTWB version {hdr['version']}")
    hdr["norm_a"] = np.array(fv[0:3], np.float32)
    hdr["norm_b"] = np.array(fv[3:6], np.float32)
    hdr["success_threshold"], hdr["ln_eps"] = fv[6], fv[7]
    tens = {}
    for i in range(hdr["n_tensors"]):
        name, code, rows, cols, _pad, off, nbytes = struct.unpack_from("<32sIIIIQQ", raw,
                                                                        256 + 64 * i)
        dt = np.dtype("<u2") if code == 1 else np.dtype("<f4")
        if code > 1 or nbytes != rows * cols * dt.itemsize or off + nbytes > len(raw):
            raise ValueError(f"VTWB tensor {name!r} malformed")
        tens[name.split(b"\0")[0].decode()] = np.frombuffer(raw, dt, rows * cols, off).reshape(
            rows, cols)
    return hdr, tens


class Model:
    def __init__(self, blob_path_or_bytes):
        if isinstance(blob_path_or_bytes, (bytes, bytearray, memoryview)):
            raw = bytes(blob_path_or_bytes)
        else:
            with open(blob_path_or_bytes, "rb") as f:
                raw = f.read()
        self.hdr, tens = parse_vtwb(raw)
        self.t = {k: (bf16_bits_to_f32(v) if v.dtype == np.uint16 else v.astype(np.float32))
                  for k, v in tens.items()}
        h = self.hdr
        self.patch, self.T, self.S, self.D = h["patch"], h["template"], h["search"], h["dim"]
        self.H, self.L, self.kpad, self.C = h["heads"], h["layers"], h["kpad"], h["head_ch"]
        self.gt, self.gs = self.T // self.patch, self.S // self.patch
        self.nt, self.ns = self.gt ** 2, self.gs ** 2
        self.eps = h["ln_eps"]
        self._folds = {}

    def folds(self, l):
        """(qkv fold, fc1 fold) of layer l, built once per rounding mode"""
        key = (l, ROUNDING)
        if key not in self._folds:
            t, p = self.t, f"l{l}."
            self._folds[key] = (
                fold_layernorm(t[p + "qkv_w"], t[p + "ln1_g"], t[p + "ln1_b"], t[p + "qkv_b"]),
                fold_layernorm(t[p + "fc1_w"], t[p + "ln2_g"], t[p + "ln2_b"], t[p + "fc1_b"]))
        return self._folds[key]

    def forward(self, patches_bits, taps=False):
        """patches_bits [N, kpad] uint16 (template rows then search rows) -> dict. Taps hold the
        residual stream as the HIP path can reproduce it: hi + lo in float32."""
        t, D = self.t, self.D
        out = {}
        a = bf16_bits_to_f32(patches_bits)
        v = (a @ t["patch_w"].T + t["patch_b"] + t["pos"]).astype(np.float32)
        mean, rstd = row_stats(v, self.eps)
        hi, lo = split_residual(v)
        if taps:
            out["tokens0"] = (hi + lo).astype(np.float32)
        for l in range(self.L):
            p = f"l{l}."
            f_qkv, f_fc1 = self.folds(l)
            qkv = folded_linear(hi, mean, rstd, f_qkv)
            q = bf16r(qkv[:, :D] * QK_SCALE)
            k = bf16r(qkv[:, D:2 * D])
            vv = bf16r(qkv[:, 2 * D:])
            o = bf16r(attention(q, k, vv, self.H))
            v = ((o @ t[p + "proj_w"].T + t[p + "proj_b"]).astype(np.float32) +
                 (hi + lo).astype(np.float32)).astype(np.float32)
            mean, rstd = row_stats(v, self.eps)
            hi, lo = split_residual(v)
            u = bf16r(gelu(folded_linear(hi, mean, rstd, f_fc1)))
            v = ((u @ t[p + "fc2_w"].T + t[p + "fc2_b"]).astype(np.float32) +
                 (hi + lo).astype(np.float32)).astype(np.float32)
            mean, rstd = row_stats(v, self.eps)
            hi, lo = split_residual(v)
            if taps:
                out[f"layer{l}"] = (hi + lo).astype(np.float32)
        x = (hi + lo).astype(np.float32)
        feat = bf16r(layernorm(x[self.nt:], t["norm_g"], t["norm_b"], self.eps))
        out["feat"] = feat
        out.update(self.head(feat))
        return out

    def head(self, feat):
        t = self.t
        relu = lambda z: np.maximum(z, np.float32(0.0)).astype(np.float32)
        t0 = bf16r(relu(feat @ t["head.w0"].T + t["head.b0"]))
        t1 = bf16r(relu(im2col3x3(t0, self.gs) @ t["head.w1"].T + t["head.b1"]))
        t2 = bf16r(relu(im2col3x3(t1, self.gs) @ t["head.w2"].T + t["head.b2"]))
        t3 = bf16r(relu(im2col3x3(t2, self.gs) @ t["head.w3"].T + t["head.b3"]))
        o = (t3 @ t["head.w4"].T + t["head.b4"]).astype(np.float32)
        return {"head_t3": t3, "head_out": o}


class Result:
    def __init__(self, success, score, bbox, fbox=None, idx=-1):
        self.success, self.score, self.bbox, self.fbox, self.idx = success, score, bbox, fbox, idx

    def __repr__(self):
        return f"Result(success={self.success}, score={self.score:.4f}, bbox={self.bbox})"


class VitTrackRef:
    """≙ vit_tracker::VitTrack as the reference host uses it (src/tracker_context.rs:21,88,90)."""

    def __init__(self, weights_path, success_threshold=None):
        self.m = Model(weights_path)
        self.thr = self.m.hdr["success_threshold"] if success_threshold is None \
            else success_threshold
        self.box = None          # float32 [x, y, w, h]
        self.tpl = None          # template patch rows
        self.last = {}

    def _pre(self, frame, box, factor, size):
        m = self.m
        return preproc(frame, box, factor, size, m.patch, m.kpad, m.hdr["norm_a"],
                       m.hdr["norm_b"])

    def init(self, frame: Frame, bbox):
        """≙ tracker.init(full_image, bbox) — return value unused by the caller (:88)"""
        self.box = np.array([bbox[0], bbox[1], bbox[2], bbox[3]], np.float32)
        self.tpl = self._pre(frame, self.box, 2.0, self.m.T)

    def update(self, frame: Frame, taps=False) -> Result:
        """≙ tracker.update(full_image) -> {success, score, bbox} (:90,120)"""
        if self.box is None:
            raise RuntimeError("update before init")
        m = self.m
        geo = crop_geometry(self.box, 4.0, m.S)
        srch = self._pre(frame, self.box, 4.0, m.S)
        patches = np.concatenate([self.tpl, srch], axis=0)
        out = m.forward(patches, taps=taps)
        ho = np.ascontiguousarray(out["head_out"], np.float32)
        hann = np.ascontiguousarray(m.t["hann"].reshape(-1), np.float32)
        dec = np.zeros(6, np.float32)
        ib = np.zeros(4, np.int32)
        lib().vto_decode(_fp(ho), _fp(hann), m.gs, _fp(geo), frame.w, frame.h, _fp(dec),
                         ib.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        score = float(dec[0])
        success = bool(score >= self.thr)
        if success:
            # next crop is cut around the INTEGER box the caller sees
            self.box = ib.astype(np.float32).copy()
        if taps:
            out["patches"] = patches
            out["geo"] = geo
            self.last = out
        return Result(success, score, tuple(int(v) for v in ib), dec[1:5].copy(), int(dec[5]))
