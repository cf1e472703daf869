"""Operator-level numerics: each HIP kernel against a plain float32 NumPy reference of the same op
(bf16 operands are exact in float32, accumulation in float32). Calls go through the C ABI."""
import numpy as np
import pytest
from scipy.special import erf

from conftest import bf16_round

pytestmark = pytest.mark.gpu

QK_SCALE = np.float32(0.125 * 1.4426950408889634)   # q is stored in log2 units


def _bits(vt, x):
    return vt.weights.f32_to_bf16_bits(np.asarray(x, np.float32))


# Exact-integer tests of the X-epilogues (the GEMMs that write the residual stream): since round 6 that stream is the 3-byte
# pair hi (bf16) + lo8 * 2^-12 (numerical specification v3), which holds every multiple of 2^-12 below 16 exactly - so the
# integer operands are scaled by 2^-12 on the W / bias / addend side (a power of two: every product and sum stays exact)
SC = np.float32(2.0 ** -12)


def _rand_bf16(vt, rng, shape, scale=1.0):
    x = (rng.standard_normal(shape) * scale).astype(np.float32)
    b = _bits(vt, x)
    return b, vt.weights.bf16_bits_to_f32(b)


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (80, 128, 128), (320, 768, 768), (720, 2304, 768),
                                   (1, 64, 192), (257, 192, 3072), (2880, 768, 768)])
def test_gemm_f32(gpu, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    ab, a = _rand_bf16(gpu, rng, (M, K))
    wb, w = _rand_bf16(gpu, rng, (N, K), 0.05)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = a @ w.T + bias
    got = gpu.op_gemm_bf16(ab, wb, bias, epilogue=0)
    tol = 1e-4 * np.sqrt(K) * np.abs(ref).max() / 10 + 1e-4
    assert np.abs(got - ref).max() < tol, np.abs(got - ref).max()


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 18, 19])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_every_tile_config(gpu, cfg, epi):
    """each tile configuration (64x64 / 128x128 with ring 2..4 and K-tile depth 64, 4-6: depth 128,
    and the 256x256 8-wave kernel) on a ragged M, against float32 NumPy"""
    rng = np.random.default_rng(cfg * 10 + epi)
    M, N, K = 720 + 37, 768, (512 if cfg in (4, 5, 6) else 384)      # K-tile depth 128 there
    ab, a = _rand_bf16(gpu, rng, (M, K))
    wb, w = _rand_bf16(gpu, rng, (N, K), 0.05)
    bias = rng.standard_normal(N).astype(np.float32)
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    z = a @ w.T + bias
    got = gpu.op_gemm_bf16(ab, wb, bias, c_init=c0, epilogue=epi, cfg=cfg)
    if epi == 0:
        assert np.abs(got - z).max() < 2e-3
    elif epi == 1:
        assert np.abs(got - (c0 + z)).max() < 2e-3
    else:
        ref = 0.5 * z * (1.0 + erf(z * 0.7071067811865476))
        assert np.all(np.abs(got - ref) <= np.abs(ref) * 2 ** -8 + 2e-3)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 18, 19])
def test_qkv_every_tile_config(gpu, cfg):
    rng = np.random.default_rng(cfg)
    B, tokens, D = 2, 100, 768
    ab, a = _rand_bf16(gpu, rng, (B * tokens, D))
    wb, w = _rand_bf16(gpu, rng, (3 * D, D), 0.04)
    bias = (rng.standard_normal(3 * D) * 0.1).astype(np.float32)
    z = a @ w.T + bias
    qk, vt_ = gpu.op_qkv_bf16(ab, wb, bias, B, tokens, D, cfg=cfg)
    ref_qk = np.concatenate([z[:, :D] * QK_SCALE, z[:, D:2 * D]], axis=1)
    assert np.all(np.abs(qk - ref_qk) <= np.abs(ref_qk) * 2 ** -8 + 1e-3)
    v = z[:, 2 * D:].reshape(B, tokens, D // 64, 64).transpose(0, 2, 3, 1).reshape(-1, 64, tokens)
    assert np.all(np.abs(vt_[:, :, :tokens] - v) <= np.abs(v) * 2 ** -8 + 1e-3)


@pytest.mark.parametrize("cfg", [0, 2, 3, 5, 7])
def test_xcd_tile_order_changes_placement_only(gpu, cfg):
    """GemmArgs.tile_order (round 6): whether an XCD's contiguous run of workgroups covers row panels x all columns (1) or
    column tiles x all rows (2) decides which L2 sees which operand, never what a tile computes - both orders (forced through
    bits 8-9 of the operator hook's configuration argument) and the launcher's own rule (0: column runs for the bf16-output
    epilogues when M < N) must give the same bits, on a ragged M whose tile count is not a multiple of 8, for GELU and QKV"""
    rng = np.random.default_rng(cfg)
    M, N, K = 720 + 37, 1536, 256
    ab, _ = _rand_bf16(gpu, rng, (M, K))
    wb, _ = _rand_bf16(gpu, rng, (N, K), 0.05)
    bias = rng.standard_normal(N).astype(np.float32)
    out = [gpu.op_gemm_bf16(ab, wb, bias, epilogue=2, cfg=cfg | o) for o in (0x000, 0x100, 0x200)]
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
    B, tokens, D = 3, 100, 512
    ab, _ = _rand_bf16(gpu, rng, (B * tokens, D))
    wb, _ = _rand_bf16(gpu, rng, (3 * D, D), 0.04)
    bias = (rng.standard_normal(3 * D) * 0.1).astype(np.float32)
    outs = [gpu.op_qkv_bf16(ab, wb, bias, B, tokens, D, cfg=cfg | o, vt_perm=1) for o in (0x000, 0x100, 0x200)]
    for qk, vt_ in outs[1:]:
        assert np.array_equal(qk, outs[0][0]) and np.array_equal(vt_, outs[0][1])


def test_gemm_exact_integers_asymmetric(gpu):
    """A = identity-like selector, W asymmetric small integers: catches swapped row/col maps and
    any staging/swizzle mix-up exactly (all values are exact in bf16 and f32)."""
    M, N, K = 128, 128, 128
    a = np.zeros((M, K), np.float32)
    a[np.arange(M), np.arange(M) % K] = 1.0
    a[np.arange(M), (np.arange(M) * 5 + 3) % K] += 2.0
    w = ((np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 7) % 17 - 8).astype(np.float32)
    ref = a @ w.T
    got = gpu.op_gemm_bf16(_bits(gpu, a), _bits(gpu, w), None, epilogue=0)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (720, 768, 3072), (131, 128, 256), (1000, 768, 1024)])
def test_gemm_4wave_exact_integers(gpu, M, N, K, cfg):
    """the 4-wave kernel's tile configurations (K-tile depth 64 and 128: different LDS row length,
    swizzle and pieces per stage) on small-integer operands - every product and sum exact, so a
    mis-staged piece or swizzle slip is a wrong integer; K = 128 is one K-tile of the deep
    configurations (prologue only), K = 256 two"""
    rng = np.random.default_rng(M + N + K + cfg)
    a = rng.integers(-4, 5, size=(M, K)).astype(np.float32)
    w = rng.integers(-4, 5, size=(N, K)).astype(np.float32)
    bias = rng.integers(-8, 9, size=N).astype(np.float32)
    c0 = rng.integers(-100, 100, size=(M, N)).astype(np.float32)
    ref = a @ w.T + bias
    assert np.array_equal(gpu.op_gemm_bf16(_bits(gpu, a), _bits(gpu, w * SC), bias * SC, epilogue=0, cfg=cfg), ref * SC)
    assert np.array_equal(gpu.op_gemm_bf16(_bits(gpu, a), _bits(gpu, w * SC), bias * SC, c_init=c0 * SC, epilogue=1,
                                           cfg=cfg), (ref + c0) * SC)
    got = gpu.op_gemm_bf16(_bits(gpu, a), _bits(gpu, w), bias, epilogue=3, cfg=cfg)
    assert np.array_equal(got, bf16_round(np.maximum(ref, 0)))


@pytest.mark.parametrize("cfg", [18])
@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (256, 512, 256), (1000, 768, 1024), (513, 256, 192)])
def test_gemm256_exact_integers(gpu, M, N, K, cfg):
    """the 256x256 8-wave kernel (config 18: 2 long phases per K-tile, one tile per workgroup) on small-integer
    operands: every product and sum is exact, so any mis-staged half-tile, swizzle slip or early
    read of a buffer shows as a wrong integer; K = 128 is the shortest supported loop (prologue +
    the two tail tiles only), K = 192 runs the steady-state body exactly once"""
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-4, 5, size=(M, K)).astype(np.float32)
    w = rng.integers(-4, 5, size=(N, K)).astype(np.float32)
    bias = rng.integers(-8, 9, size=N).astype(np.float32)
    c0 = rng.integers(-100, 100, size=(M, N)).astype(np.float32)
    ref = a @ w.T + bias
    assert np.array_equal(gpu.op_gemm_bf16(_bits(gpu, a), _bits(gpu, w * SC), bias * SC, epilogue=0, cfg=cfg), ref * SC)
    assert np.array_equal(gpu.op_gemm_bf16(_bits(gpu, a), _bits(gpu, w * SC), bias * SC, c_init=c0 * SC, epilogue=1,
                                           cfg=cfg), (ref + c0) * SC)
    got = gpu.op_gemm_bf16(_bits(gpu, a), _bits(gpu, w), bias, epilogue=3, cfg=cfg)
    assert np.array_equal(got, bf16_round(np.maximum(ref, 0)))


@pytest.mark.parametrize("cfg", [18, 19])
def test_gemm256_full_chip_exact_integers(gpu, cfg):
    """configs 18 / 19 at the bench's size (30 streams: M = 21,600, fc1 shape, 1020 workgroups = 4 rounds
    on 256 CUs) with small-integer operands: every output must be the exact integer, three launches
    in a row. A half-tile read before its LDS-DMA landed, or overwritten while still being read,
    only shows under full-chip memory load - this is the case the small shapes cannot reach."""
    rng = np.random.default_rng(2026)
    M, N, K = 21600, 3072, 768
    a = rng.integers(-4, 5, size=(M, K)).astype(np.float32)
    w = rng.integers(-4, 5, size=(N, K)).astype(np.float32)
    bias = rng.integers(-8, 9, size=N).astype(np.float32)
    ref = a @ w.T + bias          # |sums| <= 12,296: exact in float32 in any order
    ab, wb, wbs = _bits(gpu, a), _bits(gpu, w), _bits(gpu, w * SC)
    for _ in range(3):
        got = gpu.op_gemm_bf16(ab, wbs, bias * SC, epilogue=0, cfg=cfg)
        assert np.array_equal(got, ref * SC)
    c0 = rng.integers(-100, 100, size=(M, N)).astype(np.float32)
    assert np.array_equal(gpu.op_gemm_bf16(ab, wbs, bias * SC, c_init=c0 * SC, epilogue=1, cfg=cfg), (ref + c0) * SC)
    assert np.array_equal(gpu.op_gemm_bf16(ab, wb, bias, epilogue=3, cfg=cfg), bf16_round(np.maximum(ref, 0)))


@pytest.mark.parametrize("cfg", [18])
def test_gemm256_long_k_repeatable(gpu, cfg):
    """config 18 on the fc2 shape of 4 streams (48 K-tiles, 36 workgroups), five launches: the
    results must agree with float32 NumPy and be bit-identical from launch to launch (a race
    between the LDS-DMA ring and the fragment reads would show as run-to-run differences)"""
    rng = np.random.default_rng(17)
    M, N, K = 2880, 768, 3072
    ab, a = _rand_bf16(gpu, rng, (M, K))
    wb, w = _rand_bf16(gpu, rng, (N, K), 0.02)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = a @ w.T + bias
    first = gpu.op_gemm_bf16(ab, wb, bias, epilogue=0, cfg=cfg)
    assert np.abs(first - ref).max() < 2e-3
    for _ in range(4):
        assert np.array_equal(gpu.op_gemm_bf16(ab, wb, bias, epilogue=0, cfg=cfg), first)


def test_gemm_residual(gpu):
    rng = np.random.default_rng(5)
    M, N, K = 320, 768, 3072
    ab, a = _rand_bf16(gpu, rng, (M, K))
    wb, w = _rand_bf16(gpu, rng, (N, K), 0.02)
    bias = rng.standard_normal(N).astype(np.float32)
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    ref = c0 + a @ w.T + bias
    got = gpu.op_gemm_bf16(ab, wb, bias, c_init=c0, epilogue=1)
    assert np.abs(got - ref).max() < 2e-3


@pytest.mark.parametrize("epi", [2, 3])
def test_gemm_activation_bf16(gpu, epi):
    rng = np.random.default_rng(11 + epi)
    M, N, K = 336, 512, 768
    ab, a = _rand_bf16(gpu, rng, (M, K))
    wb, w = _rand_bf16(gpu, rng, (N, K), 0.05)
    bias = rng.standard_normal(N).astype(np.float32)
    z = a @ w.T + bias
    if epi == 2:
        ref = 0.5 * z * (1.0 + erf(z * 0.7071067811865476))
    else:
        ref = np.maximum(z, 0)
    got = gpu.op_gemm_bf16(ab, wb, bias, epilogue=epi)
    # output is bf16: half an ulp (2^-9 relative) plus accumulation noise
    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2 ** -8 + 2e-3)
    assert np.array_equal(got, bf16_round(got))


@pytest.mark.parametrize("B,tokens,D", [(1, 80, 128), (2, 320, 768), (1, 720, 768)])
def test_qkv_epilogue_layout(gpu, B, tokens, D):
    rng = np.random.default_rng(B * 100 + tokens)
    M = B * tokens
    ab, a = _rand_bf16(gpu, rng, (M, D))
    wb, w = _rand_bf16(gpu, rng, (3 * D, D), 0.04)
    bias = (rng.standard_normal(3 * D) * 0.1).astype(np.float32)
    z = a @ w.T + bias
    qk, vt_ = gpu.op_qkv_bf16(ab, wb, bias, B, tokens, D)
    ref_qk = np.concatenate([z[:, :D] * QK_SCALE, z[:, D:2 * D]], axis=1)
    assert np.all(np.abs(qk - ref_qk) <= np.abs(ref_qk) * 2 ** -8 + 1e-3)
    H = D // 64
    v = z[:, 2 * D:].reshape(B, tokens, H, 64).transpose(0, 2, 3, 1).reshape(B * H, 64, tokens)
    assert np.all(np.abs(vt_[:, :, :tokens] - v) <= np.abs(v) * 2 ** -8 + 1e-3)
    assert np.all(vt_[:, :, tokens:] == 0)


def _attn_ref(q, k, v, B, N, H):
    out = np.zeros((B * N, H * 64), np.float32)
    for b in range(B):
        for h in range(H):
            sl = slice(h * 64, (h + 1) * 64)
            rows = slice(b * N, (b + 1) * N)
            s = q[rows, sl] @ k[rows, sl].T
            p = np.exp2(s - s.max(axis=1, keepdims=True))   # q arrives pre-scaled by log2(e)/8
            out[rows, sl] = (p @ v[rows, sl]) / p.sum(axis=1, keepdims=True)
    return out


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("B,N,H,scale", [(1, 80, 2, 1.0), (2, 320, 12, 1.0), (1, 720, 12, 1.0),
                                         (1, 980, 16, 0.5), (1, 33, 1, 3.0), (3, 100, 2, 1.0)])
def test_attention(gpu, B, N, H, scale, mode):
    """mode 0: key-split waves, 1: independent waves, 2: K/Vt tiles shared through LDS"""
    rng = np.random.default_rng(N + H)
    D = H * 64
    qb, q = _rand_bf16(gpu, rng, (B * N, D), scale * 0.35)
    kb, k = _rand_bf16(gpu, rng, (B * N, D), scale)
    vb, v = _rand_bf16(gpu, rng, (B * N, D))
    ref = _attn_ref(q, k, v, B, N, H)
    got = gpu.op_attention_bf16(qb, kb, vb, B, N, H, mode=mode)
    err = np.abs(got - ref)
    # P is rounded to bf16 before the PV product and the output is bf16
    assert err.max() < 0.02 * max(1.0, np.abs(ref).max()), err.max()
    assert err.mean() < 2e-3


@pytest.mark.parametrize("B,N,H,scale", [(1, 80, 2, 1.0), (2, 320, 12, 1.0), (1, 720, 12, 1.0), (3, 96, 2, 1.0),
                                         (1, 16, 1, 3.0), (2, 1008, 4, 0.5), (5, 720, 12, 2.0),
                                         (3, 100, 2, 1.0), (2, 980, 16, 0.5), (1, 36, 1, 2.0)])
@pytest.mark.parametrize("mode", [3])
def test_attention_mode3(gpu, B, N, H, scale, mode):
    """mode 3: LDS-DMA ring, 64-key steps, Vt with the permuted key order (tokens % 4 == 0).
    Covers a single partial tile (16, 36), a tail of 16 and of 32 keys (80, 720 / 96), full tiles
    (320), more tiles than ring stages (1008), token counts that are not multiples of 16 (100, 980,
    36: the last 16-key group is partly padding) and a ragged last query block. Mode 3: unchecked
    first pass + careful second pass on demand."""
    rng = np.random.default_rng(N + H)
    D = H * 64
    qb, q = _rand_bf16(gpu, rng, (B * N, D), scale * 0.35)
    kb, k = _rand_bf16(gpu, rng, (B * N, D), scale)
    vb, v = _rand_bf16(gpu, rng, (B * N, D))
    ref = _attn_ref(q, k, v, B, N, H)
    got = gpu.op_attention_bf16(qb, kb, vb, B, N, H, mode=mode)
    err = np.abs(got - ref)
    assert err.max() < 0.02 * max(1.0, np.abs(ref).max()), err.max()
    assert err.mean() < 2e-3
    assert np.array_equal(got, gpu.op_attention_bf16(qb, kb, vb, B, N, H, mode=mode))   # run-to-run identical


@pytest.mark.parametrize("mode", [0, 2, 3])
def test_attention_late_maximum_rescale(gpu, mode):
    """scores that grow by far more than the lazy-max threshold / score window late in the key
    sequence: the reference must move and everything accumulated before be rescaled (mode 3: the
    unchecked first pass overflows, its row-sum test fails, the careful second pass runs)"""
    rng = np.random.default_rng(77)
    B, N, H = 1, 320, 2
    D = H * 64
    qb, q = _rand_bf16(gpu, rng, (N, D), 0.35)
    k = (rng.standard_normal((N, D)) * 1.0).astype(np.float32)
    ramp = 1.0 + 12.0 * (np.arange(N) // 64)[:, None]          # key tiles 0..4: scale 1, 13, 25, 37, 49
    kb = _bits(gpu, k * ramp)
    k = gpu.weights.bf16_bits_to_f32(kb)
    vb, v = _rand_bf16(gpu, rng, (N, D))
    ref = _attn_ref(q, k, v, B, N, H)
    got = gpu.op_attention_bf16(qb, kb, vb, B, N, H, mode=mode)
    err = np.abs(got - ref)
    assert np.isfinite(got).all()
    assert err.max() < 0.03 * max(1.0, np.abs(ref).max()), err.max()


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("level", [-20.0, -50.0, -64.0, -150.0, 55.0, 90.0])
def test_attention_uniformly_offset_scores(gpu, mode, level):
    """every score of every query sits near `level` (log2 units): inside the window the kernel uses
    p = 2^s as is (-20), below it the first step must adopt the maximum or the row would underflow
    (-64, -150), above it the reference must move before 2^s overflows (+90). Mode 3's unchecked first
    pass accepts row sums in [2^-60, 2^60]: -50 and +55 stay inside (p = 2^s as it is, far outside the
    +-32 window of the careful pass), -64 / -150 / +90 fail its test and take the second pass"""
    rng = np.random.default_rng(int(abs(level)))
    B, N, H = 1, 192, 1
    q = np.zeros((N, 64), np.float32)
    k = np.zeros((N, 64), np.float32)
    q[:, 0] = 1.0
    k[:, 0] = level                                   # common offset: q.k = level + small part
    q[:, 1:] = bf16_round(rng.standard_normal((N, 63)).astype(np.float32) * 0.25)
    k[:, 1:] = bf16_round(rng.standard_normal((N, 63)).astype(np.float32) * 0.5)
    vb, v = _rand_bf16(gpu, rng, (N, 64))
    qb, kb = _bits(gpu, q), _bits(gpu, k)
    q, k = gpu.weights.bf16_bits_to_f32(qb), gpu.weights.bf16_bits_to_f32(kb)
    ref = _attn_ref(q, k, v, B, N, H)
    got = gpu.op_attention_bf16(qb, kb, vb, B, N, H, mode=mode)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 0.02 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("cfg", [2, 3, 4, 6, 18, 19])
@pytest.mark.parametrize("tokens", [112, 100])
def test_qkv_permuted_vt_layout(gpu, cfg, tokens):
    """Vt as attention mode 3 reads it: inside every group of 16 tokens OF A STREAM the 4-token runs
    1 and 2 swap places (position = token with bits 2 and 3 exchanged). 112: streams start on
    16-token boundaries (whole 16-B pieces); 100: they do not (run-by-run placement)"""
    rng = np.random.default_rng(cfg)
    B, D = 3, 768
    ab, a = _rand_bf16(gpu, rng, (B * tokens, D))
    wb, w = _rand_bf16(gpu, rng, (3 * D, D), 0.04)
    bias = (rng.standard_normal(3 * D) * 0.1).astype(np.float32)
    z = a @ w.T + bias
    _, vt_ = gpu.op_qkv_bf16(ab, wb, bias, B, tokens, D, cfg=cfg, vt_perm=1)
    v = z[:, 2 * D:].reshape(B, tokens, D // 64, 64).transpose(0, 2, 3, 1).reshape(-1, 64, tokens)
    t = np.arange(tokens)
    pos = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    got = vt_[:, :, pos]           # got[..., t] = stored position of token t
    assert np.all(np.abs(got - v) <= np.abs(v) * 2 ** -8 + 1e-3)
    unused = np.setdiff1d(np.arange(vt_.shape[2]), pos)
    assert np.all(vt_[:, :, unused] == 0)      # padding positions stay zero


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_attention_exact_selector(gpu, mode):
    """One key dominates each query (score gap >> 1): output must equal that key's V row (exact in
    bf16) — checks the permuted k-order of the P·V product and the Vt layout with asymmetric data."""
    N, H = 96, 1
    q = np.zeros((N, 64), np.float32)
    k = np.zeros((N, 64), np.float32)
    sel = (np.arange(N) * 37 + 5) % N
    for i in range(N):     # key i <-> dims (i % 32, 32 + i // 32): disjoint ranges, no accidental ties
        k[i, i % 32] = 8.0
        k[i, 32 + i // 32] = 8.0
    for i in range(N):
        j = sel[i]
        q[i, j % 32] = 4.0
        q[i, 32 + j // 32] = 4.0
    v = ((np.arange(N)[:, None] * 5 + np.arange(64)[None, :] * 3) % 31 - 15).astype(np.float32)
    got = gpu.op_attention_bf16(_bits(gpu, q), _bits(gpu, k), _bits(gpu, v), 1, N, H, mode=mode)
    assert np.abs(got - v[sel]).max() < 1e-3


@pytest.mark.parametrize("M,D", [(80, 128), (720, 768), (100, 1024)])
def test_layernorm(gpu, M, D):
    rng = np.random.default_rng(M + D)
    x = (rng.standard_normal((M, D)) * 2 + 0.5).astype(np.float32)
    g = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = (0.1 * rng.standard_normal(D)).astype(np.float32)
    mean = x.mean(axis=1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=1, keepdims=True)
    ref = (x - mean) / np.sqrt(var + 1e-6) * g + b
    got = gpu.op_layernorm(x, g, b)
    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2 ** -8 + 1e-4)


@pytest.mark.parametrize("B,grid,C,N,cfg", [(1, 24, 128, 128, 2), (3, 24, 128, 128, 3), (2, 8, 64, 64, 0), (2, 24, 128, 128, 4),
                                             (30, 24, 128, 128, -1), (1, 28, 128, 128, 1), (5, 5, 64, 128, 2),
                                             (2, 24, 128, 128, 7), (1, 24, 128, 128, 8)])
def test_head_conv3x3_as_implicit_gemm(gpu, B, grid, C, N, cfg):
    """the head's 3x3 convolutions gather their im2col rows inside the GEMM's A loads (no im2col
    kernel, no column buffer): exact small integers against a direct NumPy convolution with zero
    padding, every tile configuration of the 4-wave kernel, maps that do not fill the last row tile"""
    rng = np.random.default_rng(B * 1000 + grid * 10 + C + N)
    t = rng.integers(-3, 4, size=(B, grid, grid, C)).astype(np.float32)
    w = rng.integers(-2, 3, size=(N, 3, 3, C)).astype(np.float32)
    bias = rng.integers(-4, 5, size=N).astype(np.float32)
    pad = np.zeros((B, grid + 2, grid + 2, C), np.float32)
    pad[:, 1:-1, 1:-1] = t
    ref = np.zeros((B, grid, grid, N), np.float64)
    for ky in range(3):
        for kx in range(3):
            ref += pad[:, ky:ky + grid, kx:kx + grid].reshape(-1, C).astype(np.float64).dot(
                w[:, ky, kx].T.astype(np.float64)).reshape(B, grid, grid, N)
    ref = np.maximum(ref + bias, 0).reshape(-1, N)
    got = gpu.op_conv3x3_relu(gpu.weights.f32_to_bf16_bits(t.reshape(-1, C)),
                              gpu.weights.f32_to_bf16_bits(w.reshape(N, 9 * C)), bias, B, grid, cfg=cfg)
    expect = gpu.weights.bf16_bits_to_f32(gpu.weights.f32_to_bf16_bits(ref.astype(np.float32)))
    assert np.array_equal(got, expect)


def _conv3x3_ref(t, w, bias):
    B, grid, _, C = t.shape
    N = w.shape[0]
    pad = np.zeros((B, grid + 2, grid + 2, C), np.float32)
    pad[:, 1:-1, 1:-1] = t
    ref = np.zeros((B, grid, grid, N), np.float64)
    for ky in range(3):
        for kx in range(3):
            ref += pad[:, ky:ky + grid, kx:kx + grid].reshape(-1, C).astype(np.float64).dot(
                w[:, ky, kx].T.astype(np.float64)).reshape(B, grid, grid, N)
    return np.maximum(ref + bias, 0).reshape(-1, N)


@pytest.mark.parametrize("B,grid,C,R,ncb", [(1, 24, 128, 0, 0), (1, 24, 128, 1, 1), (1, 24, 128, 1, 2), (2, 24, 128, 3, 2),
                                            (3, 24, 128, 4, 2), (2, 24, 128, 4, 1), (30, 24, 128, 0, 0), (2, 16, 128, 4, 2),
                                            (2, 16, 128, 7, 1), (2, 28, 128, 4, 2), (1, 28, 128, 3, 2), (3, 8, 64, 0, 0),
                                            (2, 8, 64, 3, 1), (1, 8, 64, 8, 1), (2, 5, 64, 2, 1)])
def test_head_band_kernel_conv3x3(gpu, B, grid, C, R, ncb):
    """k_head.hip, 3x3 layers: the band's input cells + halo resident in LDS, nine taps as shifted views. Exact small
    integers against a direct NumPy convolution with zero padding - every band height incl. ragged last bands (24 = 5 *
    4 + 4, 16 = 2 * 7 + 2, 5 = 2 * 2 + 1), half and full column groups, maps whose bands do not fill their last block of 16 cells,
    the launcher's own plan at one stream and at the benchmark's 30; and on random bf16 data bit-identical to the
    implicit-GEMM kernel (the same accumulation chain over k = tap * C + c)."""
    rng = np.random.default_rng(B * 1000 + grid * 10 + C + R)
    t = rng.integers(-3, 4, size=(B, grid, grid, C)).astype(np.float32)
    w = rng.integers(-2, 3, size=(C, 3, 3, C)).astype(np.float32)
    bias = rng.integers(-4, 5, size=C).astype(np.float32)
    tb, wb = gpu.weights.f32_to_bf16_bits(t.reshape(-1, C)), gpu.weights.f32_to_bf16_bits(w.reshape(C, 9 * C))
    expect = gpu.weights.bf16_bits_to_f32(gpu.weights.f32_to_bf16_bits(_conv3x3_ref(t, w, bias).astype(np.float32)))
    for _ in range(2):
        assert np.array_equal(gpu.op_headconv(tb, wb, bias, B, grid, True, R, ncb), expect)
    tr, _ = _rand_bf16(gpu, rng, (B * grid * grid, C))
    wr, _ = _rand_bf16(gpu, rng, (C, 9 * C), 0.05)
    br = rng.standard_normal(C).astype(np.float32)
    assert np.array_equal(gpu.op_headconv(tr, wr, br, B, grid, True, R, ncb),
                          gpu.op_conv3x3_relu(tr, wr, br, B, grid, cfg=2))


@pytest.mark.parametrize("B,grid,K,N,R,ncb", [(1, 24, 768, 128, 0, 0), (2, 24, 768, 128, 3, 2), (2, 24, 768, 128, 1, 1),
                                              (30, 24, 768, 128, 0, 0), (2, 28, 1024, 128, 4, 2), (2, 16, 768, 128, 5, 1),
                                              (3, 8, 128, 64, 0, 0), (2, 8, 128, 64, 3, 1), (2, 24, 64, 128, 2, 2)])
def test_head_band_kernel_conv1x1(gpu, B, grid, K, N, R, ncb):
    """k_head.hip, the 1x1 layer (K = D): A rows of the band stream through the ring beside the weights (K-tile depth 64,
    odd numbers of row blocks padded to even). Exact integers, and random data bit-identical to the 4-wave GEMM kernel's
    ReLU epilogue."""
    rng = np.random.default_rng(B + grid + K + N + R)
    M = B * grid * grid
    a = rng.integers(-3, 4, size=(M, K)).astype(np.float32)
    w = rng.integers(-3, 4, size=(N, K)).astype(np.float32)
    bias = rng.integers(-8, 9, size=N).astype(np.float32)
    ref = bf16_round(np.maximum(a @ w.T + bias, 0))
    for _ in range(2):
        assert np.array_equal(gpu.op_headconv(_bits(gpu, a), _bits(gpu, w), bias, B, grid, False, R, ncb), ref)
    ar, _ = _rand_bf16(gpu, rng, (M, K))
    wr, _ = _rand_bf16(gpu, rng, (N, K), 0.05)
    br = rng.standard_normal(N).astype(np.float32)
    assert np.array_equal(gpu.op_headconv(ar, wr, br, B, grid, False, R, ncb),
                          gpu.op_gemm_bf16(ar, wr, br, epilogue=3, cfg=2))


@pytest.mark.parametrize("B,grid,D,N,ntok,off,R,ncb", [
    (1, 24, 768, 128, 720, 144, 0, 0), (30, 24, 768, 128, 720, 144, 0, 0), (2, 24, 768, 128, 720, 144, 3, 2),
    (2, 24, 768, 128, 720, 144, 2, 1), (3, 24, 768, 64, 600, 24, 1, 1), (3, 16, 768, 128, 320, 64, 0, 0),
    (2, 16, 768, 128, 320, 64, 4, 2), (2, 15, 768, 128, 230, 5, 2, 2), (2, 15, 768, 128, 230, 5, 3, 1),
    (2, 28, 1024, 128, 980, 196, 0, 0), (2, 28, 1024, 128, 980, 196, 2, 2), (9, 28, 1024, 128, 980, 196, 1, 1)])
def test_head_band_kernel_with_the_final_layernorm_inside(gpu, B, grid, D, N, ntok, off, R, ncb):
    """k_head.hip, LNC: the 1x1 layer's workgroup normalises its band's rows of the split residual stream itself (resident
    A image, weights through the ring). Bit-identical to the LayerNorm kernel followed by the band kernel - the same
    row arithmetic by construction (ln_row, vt_common.hpp) - on every plan incl. a short last band (grid 15) and an A
    image that is not a whole number of KiB; and within bf16 rounding of a float64 LayerNorm + product."""
    rng = np.random.default_rng(B * 7 + grid + D + N + R + ncb)
    x = (rng.standard_normal((B * ntok, D)) * rng.uniform(0.2, 3.0, size=(B * ntok, 1)) +
         rng.standard_normal((B * ntok, 1))).astype(np.float32)
    xh = _bits(gpu, x)
    xhf = gpu.weights.bf16_bits_to_f32(xh)
    xl = np.clip(np.rint((x - xhf).astype(np.float32) * np.float32(4096.0)), -127, 127).astype(np.int8)      # the pair's lo8 bytes
    xlf = xl.astype(np.float32) * SC
    gamma = (1.0 + 0.2 * rng.standard_normal(D)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(D)).astype(np.float32)
    wb, w = _rand_bf16(gpu, rng, (N, D), 0.05)
    bias = (0.3 * rng.standard_normal(N)).astype(np.float32)
    two = gpu.op_headconv_ln(xh, xl, gamma, beta, wb, bias, B, grid, ntok, off, fused=False, R=R, ncb=ncb)
    for _ in range(2):
        one = gpu.op_headconv_ln(xh, xl, gamma, beta, wb, bias, B, grid, ntok, off, fused=True, R=R, ncb=ncb)
        assert np.array_equal(one, two)
    ns = grid * grid
    rows = (np.arange(B)[:, None] * ntok + off + np.arange(ns)[None, :]).ravel()
    xs = xhf[rows].astype(np.float64) + xlf[rows]
    mu = xs.mean(1, keepdims=True)
    var = ((xs - mu) ** 2).mean(1, keepdims=True)
    y = bf16_round(((xs - mu) / np.sqrt(var + 1e-6) * gamma + beta).astype(np.float32))
    ref = np.maximum(y.astype(np.float64) @ w.astype(np.float64).T + bias, 0)
    # a LayerNorm output one bf16 step off (float32 vs float64 at a rounding boundary) moves a sum of D products by < 1e-2
    assert np.all(np.abs(one - ref) <= np.abs(ref) * 2 ** -8 + 2e-2)


@pytest.mark.parametrize("tokens,B", [(720, 30), (980, 9), (100, 70)])
def test_gemm256_persistent_qkv_and_activation_full_chip(gpu, tokens, B):
    """config 19 (persistent workgroups, wave-private epilogue, next tile's prologue in flight under
    the epilogue) on more tiles than CUs with a ragged last row panel: the QKV epilogue (q/k row-major
    through the wave-private path, V transposed through the ring, permuted key order) and the ReLU
    epilogue on exact small integers; every element is compared, three launches each (a prologue
    piece overwritten or read too early only shows under full-chip load)."""
    rng = np.random.default_rng(tokens + B)
    D = 768
    M = tokens * B
    a = rng.integers(-3, 4, size=(M, D)).astype(np.float32)
    w = rng.integers(-3, 4, size=(3 * D, D)).astype(np.float32)
    bias = rng.integers(-8, 9, size=3 * D).astype(np.float32)
    z = a @ w.T + bias                      # exact integers, |z| < 7000
    ab, wb = _bits(gpu, a), _bits(gpu, w)
    ref_qk = np.concatenate([bf16_round(z[:, :D] * QK_SCALE), bf16_round(z[:, D:2 * D])], axis=1)
    v = bf16_round(z[:, 2 * D:]).reshape(B, tokens, D // 64, 64).transpose(0, 2, 3, 1).reshape(-1, 64, tokens)
    t = np.arange(tokens)
    pos = (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)
    for _ in range(3):
        qk, vt_ = gpu.op_qkv_bf16(ab, wb, bias, B, tokens, D, cfg=19, vt_perm=1)
        assert np.array_equal(qk, ref_qk)
        assert np.array_equal(vt_[:, :, pos], v)
    ref = bf16_round(np.maximum(z, 0))
    for _ in range(3):
        assert np.array_equal(gpu.op_gemm_bf16(ab, wb, bias, epilogue=3, cfg=19), ref)


# ---- the split residual stream (numerical spec v3: 3-byte pair) and the folded LayerNorm -------

def _split_pair(x):
    """float32 -> hi + lo8 * 2^-12 as the engine stores the residual stream (numerical specification v3: hi = bf16(x),
    lo8 = clamp(rint((x - hi) * 2^12), -127, 127) as a signed byte)"""
    x = np.asarray(x, np.float32)
    hi = bf16_round(x)
    lo8 = np.clip(np.rint((x - hi).astype(np.float32) * np.float32(4096.0)), -127, 127).astype(np.float32)
    return (hi + lo8 * SC).astype(np.float32)


def _row_terms(x, eps=1e-6):
    x = x.astype(np.float64)
    mean = x.mean(axis=1)
    rstd = 1.0 / np.sqrt(x.var(axis=1) + eps)
    return np.stack([rstd, -mean * rstd], axis=1)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 18])
@pytest.mark.parametrize("epi", [0, 1, 4])
def test_x_epilogues_pair_and_row_statistics(gpu, cfg, epi):
    """the three epilogues that write the residual stream (0 plain, 1 += old pair, 4 += positional rows):
    the value comes back as the 3-byte pair of the float32 result, and the finalized row terms of the
    LayerNorm that follows equal NumPy's (float64) on that float32 result - on a ragged M, every tile
    configuration, a row mean that is NOT small against the spread (chunk-wise combination must not
    cancel)"""
    rng = np.random.default_rng(cfg * 10 + epi)
    M, N, K = 720 + 37, 768, (512 if cfg in (4, 5, 6) else 384)
    ab, a = _rand_bf16(gpu, rng, (M, K))
    wb, w = _rand_bf16(gpu, rng, (N, K), 0.05)
    bias = (rng.standard_normal(N) + 3.0).astype(np.float32)          # mean ~ 3, spread ~ 1.5
    c0 = rng.standard_normal((M, N)).astype(np.float32)
    z = a @ w.T + bias
    v = z if epi == 0 else (z + (_split_pair(c0) if epi == 1 else c0)).astype(np.float32)
    got, rs = gpu.op_gemm_bf16(ab, wb, bias, c_init=c0, epilogue=epi, cfg=cfg, want_rowstat=True)
    assert np.abs(got - v).max() < 2e-3                                # accumulation order
    assert np.all(np.abs(got - _split_pair(got)) == 0)                 # it IS a 3-byte pair
    # ... and the pair OF the float32 result: at most one quantum of the low half from the float32 reference's
    assert np.abs(got - _split_pair(v)).max() <= 2e-3 + float(SC)
    ref = _row_terms(v)
    assert np.abs(rs[:, 0] / ref[:, 0] - 1).max() < 2e-5, np.abs(rs[:, 0] / ref[:, 0] - 1).max()
    assert np.abs(rs[:, 1] - ref[:, 1]).max() < 2e-4 * np.abs(ref[:, 1]).max()


@pytest.mark.parametrize("cfg", [0, 2, 3, 5, 18])
def test_x_epilogues_exact_integers(gpu, cfg):
    """small-integer operands (scaled by 2^-12, see SC): every sum is exact and representable by the pair, so the
    X-epilogues must return the exact integers (a mis-staged row, a swapped hi / lo or a wrong addend
    row is a wrong integer), the positional variant with a period shorter than M included"""
    rng = np.random.default_rng(cfg)
    M, N, K = 517, 256, 256
    a = rng.integers(-4, 5, size=(M, K)).astype(np.float32)
    w = rng.integers(-4, 5, size=(N, K)).astype(np.float32)
    bias = rng.integers(-8, 9, size=N).astype(np.float32)
    c0 = rng.integers(-100, 100, size=(M, N)).astype(np.float32)
    ref = (a @ w.T + bias) * SC
    ab, wb, bias, c0 = _bits(gpu, a), _bits(gpu, w * SC), bias * SC, c0 * SC
    assert np.array_equal(gpu.op_gemm_bf16(ab, wb, bias, epilogue=0, cfg=cfg), ref)
    assert np.array_equal(gpu.op_gemm_bf16(ab, wb, bias, c_init=c0, epilogue=1, cfg=cfg), ref + c0)
    assert np.array_equal(gpu.op_gemm_bf16(ab, wb, bias, c_init=c0, epilogue=4, cfg=cfg), ref + c0)


@pytest.mark.parametrize("cfg", [0, 2, 5, 8, 18])
def test_x_epilogues_encode_every_float_like_the_specification(gpu, cfg):
    """the encoder of the 3-byte pair, value by value: with zero operands the X-epilogue's result is its float32 bias
    (the accumulators' initial value), so ANY float32 can be put through the kernel's split - ordinary values, bf16 ties,
    ties of the byte's own rounding, the band 8 <= |x| < 16 where the remainder reaches 128 quanta and the clamp bites,
    values far beyond 16 (saturation, never a wrap), tiny values and zeros of both signs: every one must come back as
    exactly the pair the specification defines (`_split_pair`; oracle/vit_ref.py split_residual is the same arithmetic)"""
    rng = np.random.default_rng(77 + cfg)
    M, N, K = 300, 768, 256
    Q = np.float32(SC)
    hand = np.array([1.0, 1.0 + 3 * Q, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -8 + Q, -2.5 - 2.5 * Q, 2.5 + 3.5 * Q, 0.375 * Q, -0.5 * Q,
                     1.5 * Q, 0.0, -0.0, 15.96875, -15.96875, 8.0 + 2.0 ** -5, 8.0 + 2.0 ** -5 - Q / 4, 40.0625, -40.0625,
                     1000.3, -3.0e4, 127.5 * Q, 128.5 * Q], np.float32)
    ties16 = np.float32(8.0) + (np.arange(64, dtype=np.float32) * 2 + 1) * np.float32(2.0 ** -5)
    bias = np.concatenate([hand, ties16, -ties16, rng.normal(0, 0.65, 200).astype(np.float32),
                           (rng.uniform(8, 16, 200) * rng.choice([-1, 1], 200)).astype(np.float32),
                           (rng.uniform(16, 300, 100) * rng.choice([-1, 1], 100)).astype(np.float32)])
    bias = np.concatenate([bias, rng.uniform(-4, 4, N - len(bias)).astype(np.float32)])
    zeros_a, zeros_w = np.zeros((M, K), np.float32), np.zeros((N, K), np.float32)
    want = np.broadcast_to(_split_pair(bias), (M, N))
    for epi, c0 in ((0, None), (1, np.zeros((M, N), np.float32))):
        got = gpu.op_gemm_bf16(_bits(gpu, zeros_a), _bits(gpu, zeros_w), bias, c_init=c0, epilogue=epi, cfg=cfg)
        bad = np.argwhere(got != want)
        assert bad.size == 0, (epi, bad[:4], got[tuple(bad[0])], want[tuple(bad[0])], bias[bad[0][1]])
    assert np.abs(want[0, :len(hand)] - hand)[:9].max() <= float(Q) / 2            # ordinary values: half a quantum


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 18, 19])
@pytest.mark.parametrize("epi", [2, 3])
def test_folded_layernorm_in_the_bf16_epilogues(gpu, cfg, epi):
    """y = rowstat[m][0] * acc + (rowstat[m][1] * colsum[n] + bias[n]) ahead of GELU / ReLU: integer
    operands and integer row / column terms make y an exact integer (ReLU: the bf16 of it, exactly);
    config 19 at a size that takes the persistent kernel (more tiles than CUs)"""
    rng = np.random.default_rng(cfg * 7 + epi)
    M, N, K = (300 * 71, 3072, 256) if cfg == 19 else (720 + 37, 768, 256)
    a = rng.integers(-3, 4, size=(M, K)).astype(np.float32)
    w = rng.integers(-3, 4, size=(N, K)).astype(np.float32)
    bias = rng.integers(-8, 9, size=N).astype(np.float32)
    rs = np.stack([rng.integers(1, 4, size=M), rng.integers(-2, 3, size=M)], axis=1).astype(np.float32)
    cs = rng.integers(-5, 6, size=N).astype(np.float32)
    y = rs[:, :1] * (a @ w.T) + (rs[:, 1:] * cs[None, :] + bias[None, :])
    got = gpu.op_gemm_bf16(_bits(gpu, a), _bits(gpu, w), bias, epilogue=epi, cfg=cfg, rowstat=rs, colsum=cs)
    if epi == 3:
        assert np.array_equal(got, bf16_round(np.maximum(y, 0)))
    else:
        ref = 0.5 * y * (1.0 + erf(y * 0.7071067811865476))
        assert np.all(np.abs(got - ref) <= np.abs(ref) * 2 ** -8 + 2e-3)


@pytest.mark.parametrize("cfg", [2, 3, 5, 18, 19])
def test_folded_layernorm_in_the_qkv_epilogue(gpu, cfg):
    """the same row / column terms through the QKV epilogue: q (scaled), k row-major and V transposed
    (the V tiles read the row terms per register, not per lane)"""
    rng = np.random.default_rng(cfg)
    B, tokens, D = (71, 300, 768) if cfg == 19 else (2, 100, 768)
    M = B * tokens
    a = rng.integers(-3, 4, size=(M, D)).astype(np.float32)
    w = rng.integers(-3, 4, size=(3 * D, D)).astype(np.float32)
    bias = rng.integers(-8, 9, size=3 * D).astype(np.float32)
    rs = np.stack([rng.integers(1, 3, size=M), rng.integers(-2, 3, size=M)], axis=1).astype(np.float32)
    cs = rng.integers(-5, 6, size=3 * D).astype(np.float32)
    z = rs[:, :1] * (a @ w.T) + (rs[:, 1:] * cs[None, :] + bias[None, :])      # exact integers
    qk, vt_ = gpu.op_qkv_bf16(_bits(gpu, a), _bits(gpu, w), bias, B, tokens, D, cfg=cfg, rowstat=rs, colsum=cs)
    ref_qk = np.concatenate([bf16_round(z[:, :D] * QK_SCALE), bf16_round(z[:, D:2 * D])], axis=1)
    assert np.array_equal(qk, ref_qk)
    v = bf16_round(z[:, 2 * D:]).reshape(B, tokens, D // 64, 64).transpose(0, 2, 3, 1).reshape(-1, 64, tokens)
    assert np.array_equal(vt_[:, :, :tokens], v)


def test_attention_second_pass_only_where_needed(gpu):
    """mode 3 decides per workgroup (one stream and head, 128 queries): a batch in which ONE head of ONE
    stream has scores far outside the unchecked pass's range, all others ordinary ones - every head must
    come out right (the extreme one only the careful second pass can get right: 2^score overflows float32
    there), the calm ones as close to the float32 answer as a batch without the extreme head"""
    rng = np.random.default_rng(314)
    B, N, H = 2, 320, 3
    D = H * 64
    qb, q = _rand_bf16(gpu, rng, (B * N, D), 0.35)
    k = rng.standard_normal((B * N, D)).astype(np.float32)
    k[N:, 64:128] *= 40.0                       # stream 1, head 1: |scores| up to several hundred
    kb = _bits(gpu, k)
    k = gpu.weights.bf16_bits_to_f32(kb)
    vb, v = _rand_bf16(gpu, rng, (B * N, D))
    ref = _attn_ref(q, k, v, B, N, H)
    got3 = gpu.op_attention_bf16(qb, kb, vb, B, N, H, mode=3)
    assert np.isfinite(got3).all()
    assert np.abs(got3 - ref).max() < 0.03 * max(1.0, np.abs(ref).max())
    calm = np.ones((B * N, D), bool)
    calm[N:, 64:128] = False
    assert np.abs(got3[calm] - ref[calm]).max() < 0.02 * max(1.0, np.abs(ref[calm]).max())
    assert np.array_equal(got3, gpu.op_attention_bf16(qb, kb, vb, B, N, H, mode=3))      # run-to-run identical


def test_register_only_half_wave_sum_has_the_butterflys_bits(gpu, tmp_path):
    """half_wave_sum (vt_common.hpp: v_permlane16_swap + DPP, the two row sums of the final LayerNorm) against the
    __shfl_xor butterfly it replaced, lane by lane on 4,096 waves of pseudo-random values: the same partners in the same
    order, so the same bits. Built from tools/dpp_xor_check.hip on the box (the check that caught hipcc adding the swap's
    first result to itself when the builtin gets bit-cast floats)."""
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dpp_xor_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", os.path.join(root, "tools", "dpp_xor_check.hip"), "-o", exe],
                   check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of 262144 lanes differ" in r.stdout
