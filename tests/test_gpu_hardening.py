"""Robustness of the C ABI: malformed blobs, hostile frame
descriptors, resource limits and handle aliasing must give a vt_status, never a fault or an abort
(the reference host is built with panic = "abort", /root/reference/Cargo.toml:37)."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_trackers_keep_their_own_group_views(gpu, weights_tiny):
    """vt_tracker_as_group: the view belongs to its tracker (it used to be one thread-local slot
    that the second call re-pointed)"""
    w, h = 640, 480
    sa, sb = gpu.synth.MovingSquare(w, h, 64, seed=1), gpu.synth.MovingSquare(w, h, 64, seed=2)
    ta, tb = gpu.VitTrack(weights_tiny), gpu.VitTrack(weights_tiny)
    ga = ta.as_group()
    gb = tb.as_group()
    assert ga._h.value != gb._h.value and ta.as_group()._h.value == ga._h.value
    fa, fb = gpu.NV12Frame(sa.frame_nv12(0), w, h), gpu.NV12Frame(sb.frame_nv12(0), w, h)
    ta.init(fa, gpu.BBox.new(*sa.gt_box(0)))
    tb.init(fb, gpu.BBox.new(100, 50, 40, 30))
    ra, rb = ta.update(fa), tb.update(fb)
    assert ga.read_state()["frames_done"] == 1 and gb.read_state()["frames_done"] == 1
    assert list(ga.read_state()["box"]) == [float(v) for v in ra.bbox]
    assert list(gb.read_state()["box"]) == [float(v) for v in rb.bbox] or not rb.success
    ta.update(fa)
    assert ga.read_state()["frames_done"] == 2 and gb.read_state()["frames_done"] == 1


def _blob(gpu):
    cfg = gpu.weights.get_config("tiny")
    return bytearray(gpu.weights.pack_blob(cfg, gpu.weights.generate_tensors(cfg)))


def _expect_format_error(gpu, tmp_path, raw, what):
    p = tmp_path / "bad.vtw"
    p.write_bytes(bytes(raw))
    with pytest.raises(gpu.VtError) as e:
        gpu.VitTrack.new(str(p))
    assert e.value.code in (-4, -3), (what, e.value)


def test_corrupt_blobs_are_rejected(gpu, tmp_path):
    good = _blob(gpu)
    gpu.VitTrack.new(gpu.weights.ensure_weights("tiny"))     # the pristine blob loads

    def entry_off(i):
        return 256 + 64 * i

    raw = bytearray(good)                                    # offset + nbytes wraps in uint64
    name, code, rows, cols, pad, off, nbytes = struct.unpack_from("<32sIIIIQQ", raw, entry_off(3))
    struct.pack_into("<QQ", raw, entry_off(3) + 48, (1 << 64) - 16, nbytes)
    _expect_format_error(gpu, tmp_path, raw, "wrapping offset")

    raw = bytearray(good)                                    # data inside the header / table
    struct.pack_into("<QQ", raw, entry_off(0) + 48, 0, struct.unpack_from("<Q", raw, entry_off(0) + 56)[0])
    _expect_format_error(gpu, tmp_path, raw, "offset inside the header")

    raw = bytearray(good)                                    # offset beyond the file
    struct.pack_into("<Q", raw, entry_off(1) + 48, len(raw) + 4096)
    _expect_format_error(gpu, tmp_path, raw, "offset past the end")

    for field, value, what in [(6, 1_000_000, "layers"), (4, 640, "D without a LayerNorm kernel"),
                               (2, 1 << 30, "template size"), (3, 0, "search size"),
                               (10, 1 << 28, "tensor count"), (1, 0, "patch"), (7, -64, "mlp")]:
        raw = bytearray(good)
        struct.pack_into("<i", raw, 8 + 4 * field, value)
        _expect_format_error(gpu, tmp_path, raw, what)

    _expect_format_error(gpu, tmp_path, good[: len(good) // 2], "truncated data")
    _expect_format_error(gpu, tmp_path, good[:300], "truncated table")


def test_resource_limits_give_status_codes(gpu):
    big = gpu.weights.ensure_weights("cfg5")
    with pytest.raises(gpu.VtError) as e:
        gpu.Group(big, n_streams=4096)                       # status, not abort
    assert e.value.code == -1
    with pytest.raises(gpu.VtError) as e:
        gpu.Group(big, n_streams=64, max_device_mib=1024)    # 64 ViT-L streams do not fit 1 GiB
    assert e.value.code == -8 and "MiB" in str(e.value)
    g = gpu.Group(gpu.weights.ensure_weights("tiny"), n_streams=4, max_device_mib=1024)
    assert g.streams == 4


def test_state_box_rejects_non_finite_and_absurd_values(gpu, weights_tiny):
    w, h = 640, 480
    sc = gpu.synth.MovingSquare(w, h, 64, seed=3)
    trk = gpu.VitTrack(weights_tiny)
    f = gpu.NV12Frame(sc.frame_nv12(0), w, h)
    trk.init(f, gpu.BBox.new(*sc.gt_box(0)))
    g = trk.as_group()
    for box in [(float("nan"), 0, 10, 10), (0, float("inf"), 10, 10), (0, 0, float("nan"), 10),
                (0, 0, 10, 0.5), (1e9, 0, 10, 10), (0, 0, 1e7, 10)]:
        with pytest.raises(gpu.VtError) as e:
            g.set_state_box(0, box)
        assert e.value.code == -1, box
    g.set_state_box(0, (-50.0, -20.0, 30.0, 30.0))            # partly outside the frame: fine
    assert trk.update(f) is not None


def test_window_smaller_than_the_crop_reads_black_not_out_of_bounds(gpu, weights_tiny):
    """A caller-supplied windowed device frame that does not cover the search crop: samples outside
    the stored window must read as black (and never touch memory outside the planes). Reference:
    the same frame with everything outside the window painted (Y,U,V) = (16,128,128) = RGB 0,0,0."""
    import torch
    w, h = 640, 480
    rng = np.random.default_rng(5)
    ypl = rng.integers(0, 256, (h, w), dtype=np.uint8)
    uvpl = rng.integers(0, 256, (h // 2, w), dtype=np.uint8)
    x0, y0, ww, wh = 200, 150, 120, 90            # the crop of a 64x64 box here is 256 px wide
    box = gpu.BBox.new(230, 170, 64, 64)
    yb, uvb = np.full_like(ypl, 16), np.full_like(uvpl, 128)
    yb[y0:y0 + wh, x0:x0 + ww] = ypl[y0:y0 + wh, x0:x0 + ww]
    uvb[y0 // 2:(y0 + wh) // 2, x0:x0 + ww] = uvpl[y0 // 2:(y0 + wh) // 2, x0:x0 + ww]
    full = torch.from_numpy(np.concatenate([yb.reshape(-1), uvb.reshape(-1)])).cuda()
    # the window packed tightly: nothing but the window's bytes exists behind these pointers
    wy = torch.from_numpy(np.ascontiguousarray(ypl[y0:y0 + wh, x0:x0 + ww])).cuda()
    wuv = torch.from_numpy(np.ascontiguousarray(uvpl[y0 // 2:(y0 + wh) // 2, x0:x0 + ww])).cuda()
    g_full = gpu.Group(weights_tiny, n_streams=1)
    g_win = gpu.Group(weights_tiny, n_streams=1)
    f_full = gpu.frame_nv12(full.data_ptr(), full.data_ptr() + w * h, w, h)
    f_win = gpu.CFrame(wy.data_ptr(), wuv.data_ptr(), w, h, ww, ww, gpu.PIX_NV12, x0, y0, 1, ww, wh)
    g_full.init_device(0, f_full, box)
    g_win.init_device(0, f_win, box)
    ra, rb = g_full.update_device([f_full])[0], g_win.update_device([f_win])[0]
    assert np.array_equal(g_full.read_tensor("patches"), g_win.read_tensor("patches"))
    assert ra.bbox == rb.bbox and ra.score == rb.score
    # descriptors that lie about the window are refused before any kernel runs
    for bad in [gpu.CFrame(wy.data_ptr(), wuv.data_ptr(), w, h, ww, ww, gpu.PIX_NV12, x0, y0, 1, 0, 0),
                gpu.CFrame(wy.data_ptr(), wuv.data_ptr(), w, h, ww, ww, gpu.PIX_NV12, x0, y0, 1, w, wh),
                gpu.CFrame(wy.data_ptr(), wuv.data_ptr(), w, h, ww - 2, ww, gpu.PIX_NV12, x0, y0, 1, ww, wh),
                gpu.CFrame(wy.data_ptr(), wuv.data_ptr(), w, h, ww, ww, gpu.PIX_NV12, x0 + 1, y0, 1, ww, wh)]:
        with pytest.raises(gpu.VtError) as e:
            g_win.update_device([bad])
        assert e.value.code == -1


def test_a_map_the_head_band_kernel_cannot_plan_takes_the_gemm_head(gpu, tmp_path):
    """advisor, round 5: the engine asked headconv_supported() only; for C = 128 and a map of ~84..112 cells per row the
    band kernel's planner finds no band height (halo image + weight ring exceed the CU's 160 KiB of LDS even at one map
    row), launch_headconv then refused every launch and EVERY pass failed with 'kernel launch failed'. Since round 6 the
    planner is consulted for all three layer kinds when the engine is created and such a model runs its head as implicit
    GEMMs + head_out + decode. A one-layer model with a 96 x 96 score map (search 1536, patch 16, C = 128): the pass runs,
    results are finite and the same with the band kernel switched off by hand (it IS off)."""
    w = gpu.weights
    cfg = w.ModelConfig("wide_grid96_t512_s1536", 16, 512, 1536, 128, 1, head_ch=128)
    assert cfg.grid_s == 96 and cfg.n_tokens == 10240
    path = str(tmp_path / "grid96.vtw")
    with open(path, "wb") as f:
        f.write(w.pack_blob(cfg, w.generate_tensors(cfg, use_asset=False)))
    W, H = 1920, 1080
    sc = gpu.synth.MovingSquare(W, H, 200, seed=3)
    out = []
    for band in (None, 0):
        trk = gpu.VitTrack.new(path)
        if band is not None:
            trk.as_group().set_tuning("head_band", band)
        f0 = gpu.NV12Frame(sc.frame_nv12(0), W, H)
        trk.init(f0, gpu.BBox.new(*sc.gt_box(0)))
        r = [trk.update(gpu.NV12Frame(sc.frame_nv12(t), W, H)) for t in range(1, 4)]       # before the fix: VtError -5
        assert all(np.isfinite(x.score) for x in r)
        out.append([(x.success, x.score, tuple(x.bbox)) for x in r])
        names = {p["name"] for p in trk.as_group().profile_device([_dev_frame(gpu, sc.frame_nv12(4), W, H)], iters=1)}
        assert "decode" in names and not any(n.startswith("head_conv") for n in names), names     # the GEMM head ran
    assert out[0] == out[1]


_keep = []


def _dev_frame(gpu, buf, w, h):
    import torch
    t = torch.from_numpy(buf).cuda()
    _keep.append(t)
    return gpu.frame_nv12(t.data_ptr(), t.data_ptr() + w * h, w, h)
