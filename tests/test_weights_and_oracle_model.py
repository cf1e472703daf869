"""Weight-blob format, deterministic generator, and the oracle pinned by its committed fixture.
CPU only."""
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_generator_is_deterministic_and_exact(vt):
    a = vt.weights.hash_uniform("x/y", 0, 1000, 0.5, 1.0)
    b = vt.weights.hash_uniform("x/y", 0, 1000, 0.5, 1.0)
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert not np.array_equal(a, vt.weights.hash_uniform("x/y", 1, 1000, 0.5, 1.0))
    assert not np.array_equal(a, vt.weights.hash_uniform("x/z", 0, 1000, 0.5, 1.0))
    assert 0.49 < a.min() and a.max() < 1.51 and abs(a.mean() - 1.0) < 0.05
    # pinned values: integer hash -> exact float, independent of platform math libraries. Literals
    # (not recomputed): a change of the hash, the key derivation or the float conversion changes them.
    u = vt.weights.hash_uniform("pin", 0, 4, 1.0)
    assert [float(x).hex() for x in u] == ["0x1.c82d5c0000000p-1", "-0x1.f364c00000000p-1",
                                           "-0x1.e33fcc0000000p-1", "-0x1.5e88c00000000p-2"]
    assert hashlib.sha256(vt.weights.hash_uniform("pin", 0, 4096, 1.0).tobytes()).hexdigest() == \
        "9e8ddabb1e412956868e4f170faddfcd3ec44d75169d0c1a84a962782055ee3f"


def test_blob_roundtrip_and_shapes(vt, weights_tiny):
    raw = open(weights_tiny, "rb").read()
    hdr, tens = vt.weights.parse_blob(raw)
    cfg = vt.weights.get_config("tiny")
    assert (hdr["patch"], hdr["template"], hdr["search"], hdr["dim"], hdr["layers"]) == \
        (cfg.patch, cfg.template, cfg.search, cfg.dim, cfg.layers)
    assert hdr["heads"] == cfg.dim // 64 and hdr["kpad"] % 64 == 0
    assert tens["patch_w"].shape == (cfg.dim, cfg.kpad) and tens["patch_w"].dtype == np.uint16
    assert tens["pos"].shape == (cfg.n_tokens, cfg.dim)
    assert tens["l1.qkv_w"].shape == (3 * cfg.dim, cfg.dim)
    assert tens["head.w1"].shape == (cfg.head_ch, 9 * cfg.head_ch)
    assert tens["hann"].shape == (1, cfg.n_s)
    # padded patch columns are zero
    assert not tens["patch_w"][:, cfg.k_patch:].any()
    # same config -> same bytes
    again = vt.weights.pack_blob(cfg, vt.weights.generate_tensors(cfg))
    assert again == raw


def test_flop_model_matches_baseline_md(vt):
    # 58.5 / 142.3 / 687.5 GFLOP (encoder + patch embed)
    assert vt.weights.get_config("cfg2").encoder_flops() / 1e9 == pytest.approx(58.5, abs=0.1)
    assert vt.weights.get_config("cfg3").encoder_flops() / 1e9 == pytest.approx(142.3, abs=0.1)
    assert vt.weights.get_config("cfg5").encoder_flops() / 1e9 == pytest.approx(687.5, abs=0.2)


def test_oracle_reproduces_committed_fixture(vt, oracle, weights_tiny):
    fx = np.load(os.path.join(HERE, "golden", "tiny_forward.npz"))
    assert np.array_equal(
        np.frombuffer(hashlib.sha256(open(weights_tiny, "rb").read()).digest(), np.uint8),
        fx["weights_sha256"]), "tiny weight blob changed: regenerate tests/golden (make_golden.py)"
    w, h, sq, seed, t = (int(v) for v in fx["scene"])
    sc = vt.synth.MovingSquare(w, h, sq, seed=seed)
    trk = oracle.VitTrackRef(weights_tiny)
    fr = oracle.Frame.nv12(sc.frame_nv12(t), w, h)
    trk.init(fr, tuple(int(v) for v in fx["init_box"]))
    res = trk.update(fr, taps=True)
    out = trk.last
    # integer / exact-float stage: bit for bit
    assert np.array_equal(
        np.frombuffer(hashlib.sha256(out["patches"].tobytes()).digest(), np.uint8),
        fx["patches_sha256"])
    assert np.array_equal(out["geo"], fx["geo"])
    # float stages: BLAS summation order may differ between machines
    assert np.allclose(out["tokens0"], fx["tokens0"], rtol=0, atol=1e-5)
    assert np.abs(out["layer1"] - fx["layer1"]).max() < 2e-2
    assert np.abs(out["head_out"] - fx["head_out"]).max() < 5e-2
    assert np.abs(np.array(res.bbox) - fx["bbox"]).max() <= 1
    assert abs(res.score - float(fx["score"])) < 0.02


def test_oracle_preproc_properties(vt, oracle):
    """size-independent properties of the crop stage: identity scale samples pixels exactly;
    windows outside the frame are zero padded (normalised black)."""
    cfg = vt.weights.get_config("tiny")
    na, nb = vt.weights.norm_constants()
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (200, 300, 3), dtype=np.uint8)
    fr = oracle.Frame.rgb8(img)
    # box 32x32 -> search side 128 = S: scale 1, samples land exactly on pixel centres
    box = np.array([100, 60, 32, 32], np.float32)
    geo = oracle.crop_geometry(box, 4.0, cfg.search)
    assert geo[2] == 1.0 and geo[3] == 128.0
    pm = oracle.bf16_bits_to_f32(oracle.preproc(fr, box, 4.0, cfg.search, cfg.patch, cfg.kpad, na,
                                                nb))
    x0, y0 = int(100 + 16 - 64), int(60 + 16 - 64)
    g = cfg.search // cfg.patch
    for (oy, ox) in [(0, 0), (17, 40), (127, 127), (64, 3)]:
        py, px = y0 + oy, x0 + ox
        tok, kin = (oy // 16) * g + ox // 16, (oy % 16) * 16 + ox % 16
        for c in range(3):
            v = float(img[py, px, c]) if (0 <= py < 200 and 0 <= px < 300) else 0.0
            want = oracle.bf16r(np.float32(np.float32(v) * na[c] + nb[c]))
            assert pm[tok, c * 256 + kin] == want
    # a window entirely outside the frame is all "black"
    far = np.array([5000, 5000, 32, 32], np.float32)
    pm = oracle.bf16_bits_to_f32(oracle.preproc(fr, far, 4.0, cfg.search, cfg.patch, cfg.kpad, na,
                                                nb))
    for c in range(3):
        assert np.all(pm[:, c * 256:(c + 1) * 256] == oracle.bf16r(nb[c]))


def test_recommended_streams_fill_whole_gemm_rounds(vt):
    """the batch per engine pass is chosen so that the 256x256 GEMM kernel's grids are whole rounds
    of the 256 CUs: ViT-B/16 t192/s384 -> 30 streams = 85 row tiles -> 255 / 765 / 1020 tiles"""
    assert vt.weights.recommended_streams("cfg3") == 30
    cfg = vt.weights.get_config("cfg3")
    rows = -(-30 * (cfg.n_t + cfg.n_s) // 256)
    assert [rows * c for c in (3, 9, 12)] == [255, 765, 1020]
    assert vt.weights.recommended_streams("tiny") == 1      # D = 128: the 256-wide kernel never applies
    for name in ("cfg2", "cfg5"):
        b = vt.weights.recommended_streams(name)
        assert 1 < b <= 128


def test_c_abi_recommended_streams_matches_python(vt):
    for name in ("cfg2", "cfg3", "cfg5", "tiny"):
        cfg = vt.weights.get_config(name)
        mi = vt.CModelInfo()
        mi.dim, mi.mlp_dim = cfg.dim, cfg.mlp_dim
        mi.tokens_template, mi.tokens_search = cfg.n_t, cfg.n_s
        assert vt.recommended_streams(mi) == vt.weights.recommended_streams(name)


def test_engine_plan_keeps_every_engine_off_the_tile_count_cliff(vt):
    """vt_plan_engines / weights.plan_engines: n streams of one GPU over engines (Groups); sizes sum to
    n, one engine up to the recommended batch, a full engine + the rest below twice that, then halves"""
    assert vt.weights.plan_engines("cfg3", 1) == [1]
    assert vt.weights.plan_engines("cfg3", 30) == [30]
    assert vt.weights.plan_engines("cfg3", 31) == [30, 1]
    assert vt.weights.plan_engines("cfg3", 45) == [30, 15]
    assert vt.weights.plan_engines("cfg3", 60) == [30, 30]
    assert vt.weights.plan_engines("cfg3", 61) == [31, 30]
    assert vt.weights.plan_engines("cfg3", 90) == [45, 45]
    assert vt.weights.plan_engines("tiny", 50) == [50]            # no 256-wide kernel: nothing to plan
    # no engine beyond the largest batch whose GEMM operands the 256x256 kernels can address (unsigned
    # 32-bit byte offsets: B * tokens * mlp_dim * 2 < 2^32 -> 970 streams on cfg3, 534 on cfg5)
    assert vt.weights.engine_stream_cap("cfg3") == 970 and vt.weights.engine_stream_cap("cfg5") == 534
    assert vt.weights.plan_engines("cfg3", 3000) == [750, 750, 750, 750]
    assert vt.weights.plan_engines("cfg5", 1000) == [500, 500]
    with pytest.raises(ValueError):
        vt.weights.plan_engines("cfg3", 0)
    # the Python mirror against the C ABI's planner (what bench.py and a host use), swept
    for name in ("cfg2", "cfg3", "cfg5", "tiny"):
        cfg = vt.weights.get_config(name)
        mi = vt.CModelInfo()
        mi.dim, mi.mlp_dim, mi.kpad = cfg.dim, cfg.mlp_dim, cfg.kpad
        mi.tokens_template, mi.tokens_search = cfg.n_t, cfg.n_s
        for n in list(range(1, 301)) + [533, 534, 535, 969, 970, 971, 1024, 1025, 1941, 2049, 3000, 5000]:
            got = vt.plan_engines(mi, n)
            assert got == vt.weights.plan_engines(name, n), (name, n)
            assert sum(got) == n and all(0 < g <= vt.weights.engine_stream_cap(name) for g in got)
    with pytest.raises(ValueError):
        vt.plan_engines(mi, 0)


def test_oracle_parser_is_independent_and_agrees_with_the_writer(vt):
    """oracle/vit_ref.py reads the blob with its own parser (no import of the product package);
    both readers must see the same header and tensors in what the product's writer emits."""
    import inspect
    from oracle import vit_ref
    assert "gstreamer_vit_tracker_amd" not in inspect.getsource(vit_ref).replace(
        "gstreamer-vit-tracker_amd/weights.py", "")
    cfg = vt.weights.get_config("tiny")
    raw = vt.weights.pack_blob(cfg, vt.weights.generate_tensors(cfg))
    h1, t1 = vit_ref.parse_vtwb(raw)
    h2, t2 = vt.weights.parse_blob(raw)
    for k in ("patch", "template", "search", "dim", "heads", "layers", "mlp_dim", "head_ch", "kpad",
              "n_tensors", "seed", "success_threshold", "ln_eps"):
        assert h1[k] == h2[k], k
    assert np.array_equal(h1["norm_a"], h2["norm_a"]) and np.array_equal(h1["norm_b"], h2["norm_b"])
    assert list(t1) == list(t2)
    for k in t1:
        assert t1[k].dtype == t2[k].dtype and np.array_equal(t1[k], t2[k]), k
    with pytest.raises(ValueError):
        vit_ref.parse_vtwb(b"NOTAVTWB" + raw[8:])


def test_make_traj_fbox_mode_reproduces_the_fixture_and_appends_float_boxes(vt, oracle, weights_tiny, tmp_path, monkeypatch):
    """tests/golden/make_traj.py: `traj` writes a closed-loop oracle trajectory, `fbox` re-evaluates every frame of
    an existing fixture from its recorded state, insists on reproducing the stored integer boxes / scores / cells
    exactly, and appends the float boxes the teacher-forced GPU tests assert on. Here on the tiny model."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_traj", os.path.join(root, "tests", "golden", "make_traj.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    monkeypatch.setattr(mt, "HERE", str(tmp_path))
    out = str(tmp_path / "traj_tiny_6.npz")
    mt.run("tiny", weights_tiny, 6, 4, out, verbose=False)
    with np.load(out) as z:
        a = {k: z[k] for k in z.files}
    assert a["fbox"].shape == (6, 4) and np.abs(a["fbox"][:, :2] - a["bbox"][:, :2]).max() <= 0.5 + 1e-3
    # strip fbox, let the tool put it back: identical values
    b = {k: v for k, v in a.items() if k != "fbox"}
    np.savez_compressed(out, **b)
    mt.add_fbox("traj_tiny_6.npz")
    with np.load(out) as z:
        assert np.array_equal(z["fbox"], a["fbox"]) and np.array_equal(z["bbox"], a["bbox"])
    # a fixture whose stored box no longer matches what the oracle computes is refused
    b["bbox"] = b["bbox"].copy()
    b["bbox"][3, 0] += 1
    np.savez_compressed(out, **b)
    with pytest.raises(AssertionError):
        mt.add_fbox("traj_tiny_6.npz")


def test_residual_pair_v3_known_answers_and_properties(oracle):
    """numerical specification v3: hi = bf16(x), lo8 = clamp(rint((x - hi) * 2^12), -127, 127), x' = hi +
    lo8 * 2^-12. Known answers worked out by hand from the definition (bf16 keeps 8 significant bits, ties to even), then the
    properties the kernels rely on: the reconstruction error is at most half a quantum for |x| < 8 and one quantum for |x| < 16 (the
    clamp bites next to a bf16 tie), the byte saturates (never wraps) beyond, and the pair is idempotent."""
    Q = 2.0 ** -12
    kat = [  # x, hi, lo8
        (1.0, 1.0, 0),
        (1.0 + 3 * Q, 1.0, 3),                    # ulp(bf16) at 1.0 is 2^-7 = 32 quanta: well inside +-16
        (1.0 + 2.0 ** -8 + Q, 1.0 + 2.0 ** -7, -15),       # above the midpoint: hi rounds up, the byte is negative (-16 + 1)
        (1.0 + 2.0 ** -8, 1.0, 16),               # the bf16 tie goes to the even mantissa (1.0), the byte carries the half ulp
        (-2.5 - 2.5 * Q, -2.5, -2),               # a tie of the BYTE: rint(-2.5) = -2 (to even)
        (0.375 * Q, 0.375 * Q, 0),                # tiny values: hi is exact (0.375 has two significant bits), nothing is left over
        (15.96875, 16.0, -127),                   # |x| < 16: x - hi = -2^-5 = -128 quanta ... the one value that clamps: -127
        (40.0 + 0.0625, 40.0, 127),               # ulp 0.25: remainder 256 quanta, saturates at +127 (value degrades towards bf16)
    ]
    for x, hi_w, lo_w in kat:
        hi, lo = oracle.split_residual(np.array([x], np.float32))
        assert float(hi[0]) == np.float32(hi_w) and round(float(lo[0]) / Q) == lo_w, (x, float(hi[0]), float(lo[0]) / Q)
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.normal(0, 0.65, 200000), rng.uniform(-7.99, 7.99, 200000)]).astype(np.float32)
    hi, lo = oracle.split_residual(x)
    assert np.array_equal(hi, oracle.bf16r(x)) and np.array_equal(lo / np.float32(Q), np.rint(lo / np.float32(Q)))
    err = np.abs((hi + lo).astype(np.float64) - x.astype(np.float64))
    assert err.max() <= Q / 2 * (1 + 1e-6), err.max()                    # half a quantum, nothing else
    x16 = (rng.uniform(8, 15.99, 200000) * rng.choice([-1, 1], 200000)).astype(np.float32)
    x16[:64] = np.float32(8.0) + (np.arange(64, dtype=np.float32) * 2 + 1) * np.float32(2.0 ** -5)      # the bf16 ties of [8, 12)
    h16, l16 = oracle.split_residual(x16)
    e16 = np.abs((h16 + l16).astype(np.float64) - x16.astype(np.float64))
    assert Q / 2 < e16.max() <= Q, e16.max()                             # 128 quanta clamp to 127: one quantum at most
    h2, l2 = oracle.split_residual((hi + lo).astype(np.float32))
    assert np.array_equal((h2 + l2).astype(np.float32), (hi + lo).astype(np.float32))   # a stored value is a fixed point
    big = rng.uniform(16, 3000, 20000).astype(np.float32) * rng.choice([-1, 1], 20000).astype(np.float32)
    hb, lb = oracle.split_residual(big)
    assert np.abs(lb).max() <= 127 * Q
    assert (np.abs((hb + lb) - big) <= np.abs(hb - big)).all()           # saturated, the pair is never worse than bf16 alone
