"""Independent formulation of the tracker's network with torch.nn.functional, float64, NO bf16
rounding anywhere: the textbook pre-LN transformer (layer_norm, linear, scaled_dot_product_attention,
erf gelu) and the centre head as real convolutions (conv2d, padding 1).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). Two uses:

  * cross-check of oracle/vit_ref.py (tests/test_oracle_crosscheck.py): vit_ref with its roundings
    switched off must agree with this file stage by stage. vit_ref is written in the folded /
    split-residual form the HIP kernels use and was written by the same hand as the kernels; this
    file shares nothing with it but the weight-blob parser, so a shared mistake in the GELU form,
    the softmax scale, the head's im2col order or the LayerNorm folding shows up here.
  * the arbiter of tools/arbiter.py: the un-quantised answer both bf16 implementations (oracle and
    HIP) are measured against.

Like vit_ref.py it follows this build's model specification: the reference's
network is not available (/root/reference/Cargo.toml:24, src/main.rs:25) - PARITY UNPINNED. The call
shape it stands behind is VitTrack::update (/root/reference/src/tracker_context.rs:90,120).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .vit_ref import parse_vtwb, bf16_bits_to_f32


class TorchModel:
    def __init__(self, blob_path_or_bytes, dtype=torch.float64):
        if isinstance(blob_path_or_bytes, (bytes, bytearray, memoryview)):
            raw = bytes(blob_path_or_bytes)
        else:
            with open(blob_path_or_bytes, "rb") as f:
                raw = f.read()
        self.hdr, tens = parse_vtwb(raw)
        self.dtype = dtype
        self.t = {k: torch.from_numpy((bf16_bits_to_f32(v) if v.dtype == np.uint16
                                       else v.astype(np.float32)).copy()).to(dtype)
                  for k, v in tens.items()}
        h = self.hdr
        self.D, self.H, self.L, self.C = h["dim"], h["heads"], h["layers"], h["head_ch"]
        self.gt, self.gs = h["template"] // h["patch"], h["search"] // h["patch"]
        self.nt, self.ns = self.gt ** 2, self.gs ** 2
        self.eps = float(h["ln_eps"])

    @torch.no_grad()
    def forward(self, patches_bits: np.ndarray) -> dict:
        """patches_bits [N, kpad] uint16 bf16 bit patterns -> {tokens0, layer<i>, feat, head_out}
        as float64 NumPy arrays"""
        t, D, H = self.t, self.D, self.H
        out = {}
        a = torch.from_numpy(bf16_bits_to_f32(patches_bits).copy()).to(self.dtype)
        x = F.linear(a, t["patch_w"], t["patch_b"].reshape(-1)) + t["pos"]
        out["tokens0"] = x.numpy().copy()
        n = x.shape[0]
        for l in range(self.L):
            p = f"l{l}."
            h1 = F.layer_norm(x, (D,), t[p + "ln1_g"].reshape(-1), t[p + "ln1_b"].reshape(-1), self.eps)
            qkv = F.linear(h1, t[p + "qkv_w"], t[p + "qkv_b"].reshape(-1))
            q, k, v = (z.reshape(n, H, 64).transpose(0, 1) for z in qkv.split(D, dim=1))   # [H, n, 64]
            o = F.scaled_dot_product_attention(q, k, v)       # softmax(q k^T / sqrt(64)) v
            o = o.transpose(0, 1).reshape(n, D)
            x = x + F.linear(o, t[p + "proj_w"], t[p + "proj_b"].reshape(-1))
            h2 = F.layer_norm(x, (D,), t[p + "ln2_g"].reshape(-1), t[p + "ln2_b"].reshape(-1), self.eps)
            u = F.gelu(F.linear(h2, t[p + "fc1_w"], t[p + "fc1_b"].reshape(-1)))    # exact (erf) form
            x = x + F.linear(u, t[p + "fc2_w"], t[p + "fc2_b"].reshape(-1))
            out[f"layer{l}"] = x.numpy().copy()
        feat = F.layer_norm(x[self.nt:], (D,), t["norm_g"].reshape(-1), t["norm_b"].reshape(-1), self.eps)
        out["feat"] = feat.numpy().copy()
        out["head_out"] = self.head(feat).numpy().copy()
        return out

    @torch.no_grad()
    def forward_head_only(self, patches_bits: np.ndarray) -> np.ndarray:
        """The same network without the per-stage copies: [ns, 8] head logits only (what the timed
        float32 CPU baseline of oracle/cpu_fp32.py evaluates per update)."""
        t, D, H = self.t, self.D, self.H
        a = torch.from_numpy(bf16_bits_to_f32(patches_bits).copy()).to(self.dtype)
        x = F.linear(a, t["patch_w"], t["patch_b"].reshape(-1)) + t["pos"]
        n = x.shape[0]
        for l in range(self.L):
            p = f"l{l}."
            h1 = F.layer_norm(x, (D,), t[p + "ln1_g"].reshape(-1), t[p + "ln1_b"].reshape(-1), self.eps)
            qkv = F.linear(h1, t[p + "qkv_w"], t[p + "qkv_b"].reshape(-1))
            q, k, v = (z.reshape(n, H, 64).transpose(0, 1) for z in qkv.split(D, dim=1))
            o = F.scaled_dot_product_attention(q, k, v).transpose(0, 1).reshape(n, D)
            x = x + F.linear(o, t[p + "proj_w"], t[p + "proj_b"].reshape(-1))
            h2 = F.layer_norm(x, (D,), t[p + "ln2_g"].reshape(-1), t[p + "ln2_b"].reshape(-1), self.eps)
            u = F.gelu(F.linear(h2, t[p + "fc1_w"], t[p + "fc1_b"].reshape(-1)))
            x = x + F.linear(u, t[p + "fc2_w"], t[p + "fc2_b"].reshape(-1))
        feat = F.layer_norm(x[self.nt:], (D,), t["norm_g"].reshape(-1), t["norm_b"].reshape(-1), self.eps)
        return self.head(feat).numpy().astype(np.float32)

    @torch.no_grad()
    def head(self, feat: torch.Tensor) -> torch.Tensor:
        """feat [ns, D] -> [ns, 8] logits: 1x1 conv, three 3x3 convs (zero padding 1) with ReLU, then
        the 1x1 output layer; the blob stores a 3x3 kernel as [C_out][(ky*3+kx)*C_in + c]"""
        t, g, C = self.t, self.gs, self.C
        m = feat.reshape(g, g, self.D).permute(2, 0, 1).unsqueeze(0)            # [1, D, g, g]
        m = F.relu(F.conv2d(m, t["head.w0"].reshape(C, self.D, 1, 1), t["head.b0"].reshape(-1)))
        for k in (1, 2, 3):
            w = t[f"head.w{k}"].reshape(C, 3, 3, C).permute(0, 3, 1, 2)          # [C_out, C_in, ky, kx]
            m = F.relu(F.conv2d(m, w, t[f"head.b{k}"].reshape(-1), padding=1))
        o = F.conv2d(m, t["head.w4"].reshape(8, C, 1, 1), t["head.b4"].reshape(-1))
        return o.squeeze(0).permute(1, 2, 0).reshape(g * g, 8)
