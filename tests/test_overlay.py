"""Overlay drawing: the oracle's line-by-line restatements of the reference's drawing functions
(src/nv12_convert.rs:172-343, src/drawing.rs:5-50) checked on hand-derived cases (CPU), and the GPU
overlay kernel bit-exact against them (gpu)."""
import numpy as np
import pytest

W, H = 96, 64


def _blank(v=100):
    return np.full(W * H * 3 // 2, v, np.uint8)


def _y(buf):
    return buf[: W * H].reshape(H, W)


def test_oracle_rect_and_quirks(oracle):
    y = _y(oracle.draw(_blank(), W, H, [(2, 10, 8, 20, 12, 2, 250, "")]))
    # rows y1, y1+1 and y2, y2-1 for x in x1..=x2 (x2 = x + w inclusive: nv12_convert.rs:184-200)
    assert (y[8:10, 10:31] == 250).all() and (y[19:21, 10:31] == 250).all()
    assert (y[8:21, 10:12] == 250).all() and (y[8:21, 29:31] == 250).all()
    assert y[12, 15] == 100 and y[7, 10] == 100 and y[21, 10] == 100 and y[8, 31] == 100
    assert int((y == 250).sum()) == 4 * 21 + 4 * 13 - 16   # 4 full rows + 4 columns of 13 - overlap
    # (x + w) negative as usize wraps -> clamped to the last column (nv12_convert.rs:185)
    y = _y(oracle.draw(_blank(), W, H, [(2, -50, 5, 10, 6, 1, 9, "")]))
    assert (y[5, :] == 9).all() and (y[11, :] == 9).all() and (y[5:12, 0] == 9).all() and (y[5:12, W - 1] == 9).all()
    # entirely right of the frame: x1 > x2 so no horizontal edge, but the right-edge loop still
    # draws at the clamped x2 = W-1 (nv12_convert.rs:207-209) - a quirk, kept
    y = _y(oracle.draw(_blank(), W, H, [(2, W + 5, 5, 10, 6, 3, 9, "")]))
    assert (y[5:12, W - 3:W] == 9).all() and int((y == 9).sum()) == 7 * 3


def test_oracle_background_text_crosshair(oracle):
    y = _y(oracle.draw(_blank(200), W, H, [(0, 4, 4, 20, 10, 0, 150, "")]))
    assert (y[4:14, 4:24] == (200 * 105) // 255).all() and y[3, 4] == 200 and y[14, 4] == 200
    # '1' glyph rows: 00100 01100 00100 ... 01110 at scale 2, brightness 255
    y = _y(oracle.draw(_blank(0), W, H, [(1, 10, 6, 0, 0, 2, 255, "1")]))
    assert (y[6:8, 14:16] == 255).all() and y[6, 12] == 0       # row 0: only col 2
    assert (y[8:10, 12:16] == 255).all() and y[8, 10] == 0      # row 1: cols 1,2
    assert (y[18:20, 12:18] == 255).all()                       # row 6: cols 1,2,3
    # unknown characters draw nothing but advance the cursor (nv12_convert.rs:302,320)
    a = oracle.draw(_blank(0), W, H, [(1, 0, 0, 0, 0, 1, 255, "?1")])
    b = oracle.draw(_blank(0), W, H, [(1, 6, 0, 0, 0, 1, 255, "1")])
    assert np.array_equal(a, b)
    y = _y(oracle.draw(_blank(), W, H, [(3, 40, 30, 0, 0, 5, 7, "")]))
    assert (y[30, 35:46] == 7).all() and (y[25:36, 40] == 7).all() and int((y == 7).sum()) == 21


def test_oracle_cursor_and_selection(oracle):
    y = _y(oracle.draw(_blank(), W, H, [(4, 48, 32, 0, 0, 0, 0, "")]))
    assert (y[32, 23:43] == 255).all() and (y[32, 43:54] == 100).all() and (y[32, 54:74] == 255).all()
    assert (y[7:27, 48] == 255).all() and (y[27:38, 48] == 100).all()
    y = _y(oracle.draw(_blank(), W, H, [(5, 12, 6, 40, 30, 0, 0, "")]))
    assert y[6, 12] == 255 and y[6, 18] == 100 and y[6, 24] == 255      # (x / 6) % 2 == 0 dashes
    assert y[30, 12] == 255 and y[12, 12] == 255 and y[7, 12] == 100


def _random_cmds(rng, n):
    cmds = []
    glyphs = "0123456789.:- FPSTRACKINGELOD%scoremtknv?x"
    for _ in range(n):
        k = int(rng.integers(0, 6))
        x, y = int(rng.integers(-30, W + 30)), int(rng.integers(-30, H + 30))
        w, h = int(rng.integers(-10, 80)), int(rng.integers(-10, 60))
        if k in (0, 1):
            x, y, w, h = abs(x), abs(y), abs(w), abs(h)          # usize arguments in the reference
        p = int(rng.integers(0, 4)) if k != 3 else int(rng.integers(0, 30))
        text = "".join(rng.choice(list(glyphs), int(rng.integers(0, 12))))
        cmds.append((k, x, y, w, h, p, int(rng.integers(0, 256)), text))
    return cmds


@pytest.mark.gpu
def test_gpu_overlay_bit_exact_random_command_lists(gpu, oracle):
    rng = np.random.default_rng(42)
    for trial in range(30):
        base = rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)
        cmds = _random_cmds(rng, int(rng.integers(1, 12)))
        want = oracle.draw(base, W, H, cmds)
        got = gpu.overlay_nv12(base, W, H, [gpu.draw_cmd(*c) for c in cmds])
        assert np.array_equal(got, want), (trial, cmds)
        assert np.array_equal(got[W * H:], base[W * H:])          # chroma untouched


@pytest.mark.gpu
def test_gpu_overlay_probe_frame_1080p(gpu, oracle):
    """the overlay sequence of the reference's probe (src/pipeline.rs:125-174) on a 1080p frame"""
    w, h = 1920, 1080
    sc = gpu.synth.MovingSquare(w, h, 64, seed=1)
    base = sc.frame_nv12(4)
    bx, by, bw, bh = sc.gt_box(4)
    cmds = [(0, 10, 10, 400, 80, 0, 150, ""), (1, 15, 15, 0, 0, 2, 255, "TRACKING"),
            (1, 15, 40, 0, 0, 2, 255, "FPS: 60"), (1, 15, 65, 0, 0, 1, 200, "conv:0.4ms trk:1.0ms"),
            (1, 250, 15, 0, 0, 2, 255, "score: 87%"), (2, bx, by, bw, bh, 3, 255, ""),
            (3, bx + bw // 2, by + bh // 2, 0, 0, 15, 255, "")]
    want = oracle.draw(base, w, h, cmds)
    got = gpu.overlay_nv12(base, w, h, [gpu.draw_cmd(*c) for c in cmds])
    assert np.array_equal(got, want)
    assert not np.array_equal(got, base)


def test_oracle_rgb_variants(oracle):
    img = np.full((H, W, 3), 100, np.uint8)
    out = oracle.draw_rgb(img, [(2, 10, 8, 20, 12, 2, 0xFF8000, "")])
    assert (out[8:10, 10:30] == (255, 128, 0)).all() and (out[18:20, 10:30] == (255, 128, 0)).all()
    assert (out[8:20, 10:12] == (255, 128, 0)).all() and (out[8:20, 28:30] == (255, 128, 0)).all()
    assert (out[12, 15] == 100).all() and (out[8, 30] == 100).all()       # x + rw is exclusive here
    out = oracle.draw_rgb(img, [(0, -5, -5, 20, 10, 0, 0, "")])
    assert (out[0:5, 0:15] == 30).all() and (out[5, 0] == 100).all() and (out[0, 15] == 100).all()
    out = oracle.draw_rgb(img, [(4, 48, 32, 0, 0, 0, 0, "")])
    assert (out[32, 53:74] == (0, 255, 0)).all() and (out[32, 49:53] == 100).all() and (out[7:28, 48] == (0, 255, 0)).all()
    out = oracle.draw_rgb(img, [(5, 12, 6, 40, 30, 0, 0, "")])
    assert (out[6, 12] == (255, 255, 0)).all() and (out[6, 18] == 100).all()
    # clipped by set_pixel's bounds test, no wrap quirks in this variant
    out = oracle.draw_rgb(img, [(3, -3, 2, 0, 0, 6, 0x0000FF, "")])
    assert (out[2, 0:4] == (0, 0, 255)).all() and (out[2, 4] == 100).all()


@pytest.mark.gpu
def test_gpu_overlay_rgb_bit_exact_random_command_lists(gpu, oracle):
    rng = np.random.default_rng(7)
    for trial in range(30):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        cmds = []
        for c in _random_cmds(rng, int(rng.integers(1, 12))):
            k, x, y, w, h, p, v, text = c
            if k == 0 and (w < 0 or h < 0):
                w, h = abs(w), abs(h)            # an inverted background range panics in the reference
            cmds.append((k, x, y, w, h, p, int(rng.integers(0, 1 << 24)), text))
        want = oracle.draw_rgb(img, cmds)
        got = gpu.overlay_rgb8(img, [gpu.draw_cmd(*c) for c in cmds])
        assert np.array_equal(got, want), (trial, cmds)
