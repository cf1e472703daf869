"""Weight blob ("VTWB0001") writer/reader and the deterministic synthetic-weight generator.

The reference loads `object_tracking_vittrack_2023sep.rknn` from an absolute path on the
author's board (/root/reference/src/main.rs:25); neither that file nor its architecture is
available. The models here are therefore build-defined: an
OSTrack-style one-stream ViT encoder whose weights are a pure function of (seed, tensor name,
element index) — integer hash -> exact float32 -> bf16, so every machine generates the same
bits — plus a small convolutional centre head whose trained weights are committed under
assets/ (they were fitted against exactly these encoder weights, tests/golden/fit_head.py).

Blob layout (little endian):
  [0,256)    header: magic "VTWB0001", 20 x int32 model fields, 8 x float32
  [256, ...) tensor table, 64 B per entry: name[32], dtype u32 (0=f32, 1=bf16), rows u32,
             cols u32, pad u32, offset u64, nbytes u64
  data       each tensor 256-B aligned
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, asdict

import numpy as np

MAGIC = b"VTWB0001"
HEADER_BYTES = 256
ENTRY_BYTES = 64
ALIGN = 256
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass(frozen=True)
class ModelConfig:
    name: str
    patch: int
    template: int
    search: int
    dim: int
    layers: int
    mlp_ratio: int = 4
    head_ch: int = 128
    seed: int = 0

    @property
    def heads(self) -> int:
        return self.dim // 64

    @property
    def mlp_dim(self) -> int:
        return self.dim * self.mlp_ratio

    @property
    def grid_t(self) -> int:
        return self.template // self.patch

    @property
    def grid_s(self) -> int:
        return self.search // self.patch

    @property
    def n_t(self) -> int:
        return self.grid_t ** 2

    @property
    def n_s(self) -> int:
        return self.grid_s ** 2

    @property
    def n_tokens(self) -> int:
        return self.n_t + self.n_s

    @property
    def k_patch(self) -> int:
        return 3 * self.patch * self.patch

    @property
    def kpad(self) -> int:
        return (self.k_patch + 63) // 64 * 64

    def encoder_flops(self) -> float:
        """per layer 24*N*D^2 + 4*N^2*D, patch-embed 2*N*3p^2*D."""
        n, d = self.n_tokens, self.dim
        per_layer = 2.0 * n * d * (3 * d) + 2.0 * n * d * d + 2 * 2.0 * n * d * self.mlp_dim \
            + 4.0 * n * n * d
        return self.layers * per_layer + 2.0 * n * self.k_patch * d

    def head_flops(self) -> float:
        c = self.head_ch
        return 2.0 * self.n_s * (self.dim * c + 3 * 9 * c * c + c * 8)


# BASELINE.json configs (cfg2, cfg3, cfg5) and a tiny model for fast CPU tests
CONFIGS = {
    "vitb16_t128_s256": ModelConfig("vitb16_t128_s256", 16, 128, 256, 768, 12),
    "vitb16_t192_s384": ModelConfig("vitb16_t192_s384", 16, 192, 384, 768, 12),
    "vitl14_t196_s392": ModelConfig("vitl14_t196_s392", 14, 196, 392, 1024, 24),
    "tiny_t64_s128": ModelConfig("tiny_t64_s128", 16, 64, 128, 128, 2, head_ch=64),
}
ALIASES = {"cfg2": "vitb16_t128_s256", "cfg3": "vitb16_t192_s384", "cfg5": "vitl14_t196_s392",
           "tiny": "tiny_t64_s128"}


def get_config(name: str) -> ModelConfig:
    return CONFIGS[ALIASES.get(name, name)]


# ---- bf16 helpers -------------------------------------------------------------------------

def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit pattern (uint16), round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(
        np.float32)


# ---- deterministic generator --------------------------------------------------------------

def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for ch in s.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniform(name: str, seed: int, count: int, amp: float, offset: float = 0.0) -> np.ndarray:
    """offset + amp * u, u uniform on the 2^24 grid in [-1, 1): integer hash -> exact float32,
    one float32 multiply, one float32 add. Identical bits on every machine."""
    key = np.uint64((_fnv1a64(name) ^ (seed * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(count, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + key
        z = _splitmix64(idx)
    u24 = (z >> np.uint64(40)).astype(np.int64) - (1 << 23)
    u = (u24.astype(np.float32) / np.float32(1 << 23)).astype(np.float32)
    return (u * np.float32(amp) + np.float32(offset)).astype(np.float32)


def hann2d(grid: int) -> np.ndarray:
    """OSTrack-style centred Hann window on the score grid (float64 math, rounded once)."""
    i = np.arange(1, grid + 1, dtype=np.float64)
    w = 0.5 * (1.0 - np.cos(2.0 * np.pi * i / (grid + 1)))
    return np.outer(w, w).astype(np.float32)


def norm_constants():
    a = [np.float32(1.0 / (255.0 * s)) for s in IMAGENET_STD]
    b = [np.float32(-m / s) for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)]
    return np.array(a, np.float32), np.array(b, np.float32)


def head_tensor_shapes(cfg: ModelConfig):
    c = cfg.head_ch
    return {
        "head.w0": (c, cfg.dim), "head.b0": (1, c),
        "head.w1": (c, 9 * c), "head.b1": (1, c),
        "head.w2": (c, 9 * c), "head.b2": (1, c),
        "head.w3": (c, 9 * c), "head.b3": (1, c),
        "head.w4": (8, c), "head.b4": (1, 8),
    }


HEAD_BF16 = ("head.w0", "head.w1", "head.w2", "head.w3")


def recommended_streams(cfg: "ModelConfig | str", max_streams: int = 128, cus: int = 256) -> int:
    """Streams per engine pass that fill the MI355X's 256 CUs in whole rounds of the 256x256 GEMM
    kernel (one workgroup per CU): the encoder GEMMs have ceil(B * tokens / 256) row tiles times
    D/256, 3D/256 and 4D/256 column tiles, and a pass whose tile counts sit just above a multiple
    of 256 pays for an almost empty extra round in every GEMM (measured on ViT-B/16 t192/s384:
    30 streams 5,584 frames/s, 31 streams 4,439). Returns the smallest B <= max_streams whose worst
    GEMM wastes less than 2 % of its rounds (smallest: twice the batch was slower per frame, its
    MLP activations no longer fit the 256 MiB Infinity Cache); 1 if no such B exists or the widths
    do not fit that kernel."""
    if isinstance(cfg, str):
        cfg = get_config(cfg)
    if cfg.dim % 256:
        return 1
    tokens = cfg.n_t + cfg.n_s
    for b in range(1, max_streams + 1):
        rows = -(-b * tokens // 256)
        worst = 1.0
        for cols in (cfg.dim // 256, 3 * cfg.dim // 256, cfg.mlp_dim // 256):
            t = rows * cols
            worst = min(worst, t / (cus * -(-t // cus)))
        if worst >= 0.98:
            return b
    return 1


def head_asset_path(cfg: ModelConfig) -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets",
                        f"head_{cfg.name}.npz")


def generate_tensors(cfg: ModelConfig, head: dict | None = None, use_asset: bool = True):
    """Ordered dict name -> (dtype_code, 2-D array). dtype_code 1: uint16 bf16 bits, 0: float32."""
    s3 = float(np.sqrt(3.0))
    d, L = cfg.dim, cfg.layers
    seed = cfg.seed
    out: dict[str, tuple[int, np.ndarray]] = {}

    def bf(name, rows, cols, amp):
        out[name] = (1, f32_to_bf16_bits(hash_uniform(cfg.name + "/" + name, seed, rows * cols,
                                                      amp)).reshape(rows, cols))

    def f32(name, rows, cols, amp, offset=0.0):
        out[name] = (0, hash_uniform(cfg.name + "/" + name, seed, rows * cols, amp,
                                     offset).reshape(rows, cols))

    # patch embedding: conv p x p stride p as a GEMM, weight [D][kpad], padded columns zero
    w = hash_uniform(cfg.name + "/patch_w", seed, d * cfg.k_patch, 0.02 * s3).reshape(d,
                                                                                      cfg.k_patch)
    wp = np.zeros((d, cfg.kpad), np.float32)
    wp[:, :cfg.k_patch] = w
    out["patch_w"] = (1, f32_to_bf16_bits(wp).reshape(d, cfg.kpad))
    f32("patch_b", 1, d, 0.02)
    f32("pos", cfg.n_tokens, d, 0.02 * s3)  # rows: template tokens then search tokens
    for l in range(L):
        p = f"l{l}."
        f32(p + "ln1_g", 1, d, 0.1, 1.0)
        f32(p + "ln1_b", 1, d, 0.02)
        # q,k rows larger than v rows so that the softmax is not flat
        qk = hash_uniform(cfg.name + "/" + p + "qkv_w.qk", seed, 2 * d * d, 0.04 * s3)
        v = hash_uniform(cfg.name + "/" + p + "qkv_w.v", seed, d * d, 0.02 * s3)
        out[p + "qkv_w"] = (1, f32_to_bf16_bits(np.concatenate([qk, v])).reshape(3 * d, d))
        f32(p + "qkv_b", 1, 3 * d, 0.02)
        bf(p + "proj_w", d, d, 0.02 * s3 / np.sqrt(2.0 * L))
        f32(p + "proj_b", 1, d, 0.02)
        f32(p + "ln2_g", 1, d, 0.1, 1.0)
        f32(p + "ln2_b", 1, d, 0.02)
        bf(p + "fc1_w", cfg.mlp_dim, d, 0.02 * s3)
        f32(p + "fc1_b", 1, cfg.mlp_dim, 0.02)
        bf(p + "fc2_w", d, cfg.mlp_dim, 0.02 * s3 / np.sqrt(2.0 * L))
        f32(p + "fc2_b", 1, d, 0.02)
    f32("norm_g", 1, d, 0.1, 1.0)
    f32("norm_b", 1, d, 0.02)

    # centre head: committed trained asset if present, else seeded random (runs, does not track)
    shapes = head_tensor_shapes(cfg)
    if head is None and use_asset and os.path.exists(head_asset_path(cfg)):
        with np.load(head_asset_path(cfg)) as z:
            head = {k: z[k] for k in z.files}
    for name, (rows, cols) in shapes.items():
        if head is not None and name in head:
            arr = np.asarray(head[name], np.float32).reshape(rows, cols)
        else:
            fan_in = cols if name.startswith("head.w") else 1
            amp = float(np.sqrt(6.0 / fan_in)) if name.startswith("head.w") else 0.01
            arr = hash_uniform(cfg.name + "/" + name, seed, rows * cols, amp).reshape(rows, cols)
        if name in HEAD_BF16:
            out[name] = (1, f32_to_bf16_bits(arr).reshape(rows, cols))
        else:
            out[name] = (0, arr.astype(np.float32))
    out["hann"] = (0, hann2d(cfg.grid_s).reshape(1, cfg.n_s))
    return out


def pack_blob(cfg: ModelConfig, tensors: dict) -> bytes:
    names = list(tensors.keys())
    table_off = HEADER_BYTES
    data_off = (table_off + ENTRY_BYTES * len(names) + ALIGN - 1) // ALIGN * ALIGN
    entries, chunks, off = [], [], data_off
    for n in names:
        code, arr = tensors[n]
        arr = np.ascontiguousarray(arr)
        assert arr.ndim == 2 and len(n) < 32
        assert arr.dtype == (np.uint16 if code == 1 else np.float32), n
        raw = arr.tobytes()
        entries.append(struct.pack("<32sIIIIQQ", n.encode(), code, arr.shape[0], arr.shape[1], 0,
                                   off, len(raw)))
        padded = (len(raw) + ALIGN - 1) // ALIGN * ALIGN
        chunks.append(raw + b"\0" * (padded - len(raw)))
        off += padded
    na, nb = norm_constants()
    ints = [1, cfg.patch, cfg.template, cfg.search, cfg.dim, cfg.heads, cfg.layers, cfg.mlp_dim,
            cfg.head_ch, cfg.kpad, len(names), cfg.seed] + [0] * 8
    floats = list(na) + list(nb) + [0.20, 1e-6]  # default success threshold, LayerNorm eps
    header = MAGIC + struct.pack("<20i", *ints) + struct.pack("<8f", *floats)
    header += b"\0" * (HEADER_BYTES - len(header))
    table = b"".join(entries)
    table += b"\0" * (data_off - table_off - len(table))
    return header + table + b"".join(chunks)


def parse_blob(blob: bytes | memoryview):
    """-> (header dict, {name: float32 or uint16 array})."""
    mv = memoryview(blob)
    assert bytes(mv[:8]) == MAGIC, "bad magic"
    ints = struct.unpack_from("<20i", mv, 8)
    floats = struct.unpack_from("<8f", mv, 8 + 80)
    hdr = dict(version=ints[0], patch=ints[1], template=ints[2], search=ints[3], dim=ints[4],
               heads=ints[5], layers=ints[6], mlp_dim=ints[7], head_ch=ints[8], kpad=ints[9],
               n_tensors=ints[10], seed=ints[11], norm_a=np.array(floats[0:3], np.float32),
               norm_b=np.array(floats[3:6], np.float32), success_threshold=floats[6],
               ln_eps=floats[7])
    tensors = {}
    for i in range(hdr["n_tensors"]):
        name, code, rows, cols, _, off, nbytes = struct.unpack_from(
            "<32sIIIIQQ", mv, HEADER_BYTES + i * ENTRY_BYTES)
        name = name.split(b"\0")[0].decode()
        dt = np.uint16 if code == 1 else np.float32
        tensors[name] = np.frombuffer(mv, dtype=dt, count=rows * cols, offset=off).reshape(rows,
                                                                                          cols)
    return hdr, tensors


def default_cache_dir() -> str:
    return os.environ.get("VT_WEIGHTS_DIR", os.path.join("/tmp", "vt_weights"))


def ensure_weights(cfg_name: str, path: str | None = None, head: dict | None = None,
                   use_asset: bool = True, force: bool = False) -> str:
    """Generate (once) and return the path of the weight blob for a named config."""
    cfg = get_config(cfg_name)
    if path is None:
        os.makedirs(default_cache_dir(), exist_ok=True)
        tag = "" if use_asset else "_randhead"
        path = os.path.join(default_cache_dir(), f"{cfg.name}_seed{cfg.seed}{tag}.vtw")
    asset = head_asset_path(cfg)
    stale = (use_asset and head is None and os.path.exists(asset) and os.path.exists(path)
             and os.path.getmtime(asset) > os.path.getmtime(path))
    if force or stale or head is not None or not os.path.exists(path):
        blob = pack_blob(cfg, generate_tensors(cfg, head=head, use_asset=use_asset))
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, path)
    return path


def engine_stream_cap(cfg: "ModelConfig | str", max_streams: int = 1024) -> int:
    """largest engine for which every encoder GEMM still runs on the 256x256 kernels: they address an
    operand with unsigned 32-bit byte offsets, and the widest A operand of a pass is max(dim, mlp_dim,
    kpad) bf16 columns by B * tokens rows (mirror of engine_stream_cap in csrc/vt_engine.hip)"""
    if isinstance(cfg, str):
        cfg = get_config(cfg)
    width = max(cfg.dim, cfg.mlp_dim, cfg.kpad)
    return max(1, min(max_streams, ((1 << 32) - 1) // (cfg.n_tokens * width * 2)))


def plan_engines(cfg: "ModelConfig | str", n_streams: int) -> list:
    """Engine (Group) sizes for n_streams on one GPU (mirror of vt_plan_engines in the C ABI): one
    engine up to R = recommended_streams(cfg); an engine of R plus one with the rest below 2R (the two
    engines' kernels overlap on the chip, so the second one's nearly empty rounds cost little: ViT-B/16
    t192/s384, 31 streams: 172 us per frame as 30 + 1, 197 in one engine, 164 at 30 streams); two
    engines of n/2 from 2R on (31 + 30: 165 us; three concurrent engines measured worse than two:
    profiles/r02_engine_splits.txt); never an engine beyond engine_stream_cap. bench.py and the tests
    use the C ABI's vt_plan_engines; this mirror exists for hosts that plan before loading the library
    and is swept against it in tests/test_weights_and_oracle_model.py."""
    if n_streams < 1:
        raise ValueError("n_streams < 1")
    r = recommended_streams(cfg)
    bmax = engine_stream_cap(cfg)
    k = 1 if (r <= 1 or n_streams <= r) else 2
    while -(-n_streams // k) > bmax:
        k += 1
    if k == 2 and n_streams < 2 * r:
        return [r, n_streams - r]
    return [n_streams // k + (1 if i < n_streams % k else 0) for i in range(k)]


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="write a synthetic weight blob")
    ap.add_argument("config", choices=sorted(list(CONFIGS) + list(ALIASES)))
    ap.add_argument("-o", "--out", default=None)
    a = ap.parse_args()
    print(ensure_weights(a.config, a.out, force=True))
