"""Parity of the HIP hot path against the CPU oracle, through the C ABI, on the same seeded inputs.

Bars: integer / byte stages bit-exact (NV12->RGB8, and the patch matrix produced
by crop+resize+normalise, whose float ops are single IEEE operations in a fixed order); network
tensors within bf16 tolerances; boxes within +-1 px of the oracle's, mean IoU >= 0.99.
"""
import numpy as np
import pytest

from conftest import iou

pytestmark = pytest.mark.gpu


# ---- stage (a): the reference's colour converter, bit-exact ----------------------------------

@pytest.mark.parametrize("w,h", [(64, 48), (640, 480), (1920, 1080), (3840, 2160), (7680, 4320), (5120, 3277), (66, 50), (31, 17),
                                 (33, 16), (48, 31)])
def test_nv12_full_frame_bit_exact(gpu, oracle, w, h):
    rng = np.random.default_rng(w * h)
    n = w * h + w * ((h + 1) // 2) + 2
    buf = rng.integers(0, 256, n, dtype=np.uint8)
    ref, st = oracle.nv12_to_rgb8(buf, w, h, 4)
    assert st == 0
    got = gpu.nv12_full_to_rgb(buf, w, h)
    assert np.array_equal(got, ref)


def test_nv12_short_buffer_gives_zero_frame(gpu, oracle):
    w, h = 64, 48
    buf = np.full(w * h * 3 // 2 - 1, 200, np.uint8)
    ref, st = oracle.nv12_to_rgb8(buf, w, h, 1)
    assert st == 1 and not ref.any()
    got = gpu.nv12_full_to_rgb(buf, w, h)
    assert not got.any()


def test_nv12_full_frame_synthetic_1080p(gpu, oracle):
    sc = gpu.synth.MovingSquare(1920, 1080, 64, seed=0)
    buf = sc.frame_nv12(3)
    ref, _ = oracle.nv12_to_rgb8(buf, 1920, 1080, 8)
    assert np.array_equal(gpu.nv12_full_to_rgb(buf, 1920, 1080), ref)
    assert np.array_equal(ref, sc.frame_rgb8(3))


# ---- stage (b): crop + resize + normalise -> patch matrix, bit-exact ---------------------------

def _patches_case(gpu, oracle, weights, frame_np, oframe, box):
    trk = gpu.VitTrack.new(weights)
    ref = oracle.VitTrackRef(weights)
    trk.init(frame_np, gpu.BBox.new(*box))
    ref.init(oframe, box)
    r_gpu = trk.update(frame_np)
    r_ref = ref.update(oframe, taps=True)
    g = trk.as_group()
    mi = trk.model_info()
    got = g.read_tensor("patches").reshape(mi.tokens_template + mi.tokens_search, mi.kpad)
    want = oracle.bf16_bits_to_f32(ref.last["patches"])
    return got, want, r_gpu, r_ref


@pytest.mark.parametrize("box", [(288, 208, 64, 64), (300, 200, 41, 77), (5, 3, 50, 40),
                                 (600, 440, 30, 30), (100, 100, 333, 201), (200, 150, 80, 80), (250, 180, 100, 70)])
def test_patch_matrix_bit_exact_nv12(gpu, oracle, weights_tiny, box):
    """boxes for every path of the crop kernel at search size 128: 30 px -> the 16-KiB tile buffer, 41x77 / 50x40 / 64 px
    -> the 32-KiB tier, 80 px / 100x70 -> the 64-KiB tier, 333x201 -> per-pixel fetches (the engine picks the tier from
    the box it knows: round 5)"""
    sc = gpu.synth.MovingSquare(640, 480, 64, seed=1)
    buf = sc.frame_nv12(0)
    got, want, _, _ = _patches_case(gpu, oracle, weights_tiny, gpu.NV12Frame(buf, 640, 480),
                                    oracle.Frame.nv12(buf, 640, 480), box)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("box", [(288, 208, 64, 64), (200, 150, 80, 80), (600, 440, 30, 30), (100, 100, 333, 201)])
def test_crop_buffer_tier_changes_no_value(gpu, weights_tiny, box):
    """the crop kernel's LDS buffer tier is a choice of speed: forced to each tier (vt_group_set_tuning "crop_tier") the
    patch matrix and the result are identical to the automatic choice - also when the forced buffer is too small for
    the box (tiles then take the per-pixel path) - graph replay and eager"""
    w, h = 640, 480
    sc = gpu.synth.MovingSquare(w, h, 64, seed=1)
    f = gpu.NV12Frame(sc.frame_nv12(0), w, h)
    out = []
    for tier in (-1, 0, 1, 2):
        for use_graph in (True, False):
            g = gpu.Group(weights_tiny, n_streams=2, use_graph=use_graph)
            g.set_tuning("crop_tier", tier)
            for i in range(2):
                g.init_host(i, f, gpu.BBox.new(*box))
            r = [g.update_host([f, f]) for _ in range(2)][-1]
            out.append((g.read_tensor("patches", 1).tobytes(), [(x.bbox, x.score) for x in r]))
            del g
    assert all(o == out[0] for o in out)


def test_no_graph_capture_inside_an_update_when_a_target_grows_through_the_crop_tiers(gpu, weights_cfg3):
    """every crop tier's pass is captured and instantiated when the engine is created: a live stream
    (/root/reference/src/pipeline.rs:26-37: 60 fps) whose target grows through both tier boundaries (~130 / ~200 px at
    search 384) replays three different graphs and captures none - the capture counter stays at its creation value and no
    update near a crossing stalls (a capture + instantiate of 66 launches costs several update times)"""
    import time
    import torch
    w, h = 1920, 1080
    sc = gpu.synth.MovingSquare(w, h, 64, seed=2)
    buf = torch.from_numpy(sc.frame_nv12(0)).cuda()
    trk = gpu.VitTrack.new(weights_cfg3)
    g = trk.as_group()
    assert g.graph_captures() == 3                      # all three tiers, before the first frame
    p = buf.data_ptr()
    trk.init_nv12_device(p, p + w * h, w, h, w, w, gpu.BBox.new(900, 480, 100, 100))
    sizes = [100] * 20 + list(range(100, 282, 2))       # warm, then grow through both boundaries
    lat = []
    for s in sizes:
        g.set_state_box(0, [960 - s / 2, 540 - s / 2, s, s])
        a = time.perf_counter()
        trk.update_nv12_device(p, p + w * h, w, h, w, w)
        lat.append(time.perf_counter() - a)
    assert g.graph_captures() == 3, "a pass captured a graph on the hot path"
    rep = g.read_tensor("graph_replays")
    assert rep.sum() == len(sizes) and (rep > 0).all(), rep      # all three captured passes really ran
    steady = np.array(lat[20:])
    p50, p99 = np.median(steady), np.percentile(steady, 99)
    print(f"tier crossing: replays per tier {rep.tolist()}, update p50 {p50 * 1e3:.3f} ms, p99 {p99 * 1e3:.3f} ms, max {steady.max() * 1e3:.3f} ms")
    assert p99 <= 1.2 * p50, (p50, p99)
    # a tuning change re-captures at once (inside set_tuning), never inside the next update
    g.set_tuning("head_band", 1)
    assert g.graph_captures() == 6
    trk.update_nv12_device(p, p + w * h, w, h, w, w)
    assert g.graph_captures() == 6
    # a 30-stream engine: the same three captures at creation
    g30 = gpu.Group(weights_cfg3, n_streams=30)
    assert g30.graph_captures() == 3
    # eager engines capture nothing
    ge = gpu.Group(weights_cfg3, n_streams=1, use_graph=False)
    assert ge.graph_captures() == 0


@pytest.mark.parametrize("w,h,n", [(64, 48, 3), (1920, 1080, 5), (31, 17, 70), (66, 50, 2), (3840, 2160, 2), (64, 49, 3), (48, 31, 130),
                                   (1920, 1081, 2)])
def test_batched_converter_is_the_single_frame_converter_per_frame(gpu, oracle, w, h, n):
    """vt_nv12_to_rgb8_batch_device: n frames in one launch per 64, every frame bit-exact with the reference's
    converter (oracle restatement of /root/reference/src/nv12_convert.rs:46-169), odd sizes through the general
    kernel, a short buffer -> that frame all zero (:48-50), a buffer that does not cover an odd frame's reads refused"""
    import torch
    rng = np.random.default_rng(w * h + n)
    nbytes = (w * h + w * ((h + 1) // 2) + 2 + 255) // 256 * 256      # 16-B aligned frames: sizes with w % 16 == 0 take the wide kernel
    host = rng.integers(0, 256, (n, nbytes), dtype=np.uint8)
    src = torch.from_numpy(host).cuda()
    dst = torch.full((n, w * h * 3), 7, dtype=torch.uint8, device="cuda")
    lens = [nbytes] * n
    short = n // 2
    lens[short] = w * h * 3 // 2 - 1                     # the reference returns a zero frame for this one
    gpu.nv12_to_rgb8_batch_device([src[i].data_ptr() for i in range(n)], lens, w, h, [dst[i].data_ptr() for i in range(n)])
    torch.cuda.synchronize()
    got = dst.cpu().numpy()
    for i in range(n):
        ref, st = oracle.nv12_to_rgb8(host[i][:lens[i]], w, h, 2)
        assert st == (1 if i == short else 0)
        assert np.array_equal(got[i].reshape(h, w, 3), ref), i
    if (w & 1) or (h & 1):                               # odd frames read past w*h*3/2: a buffer that stops there is refused
        lens[0] = w * h * 3 // 2
        with pytest.raises(gpu.VtError) as e:
            gpu.nv12_to_rgb8_batch_device([src[i].data_ptr() for i in range(n)], lens, w, h, [dst[i].data_ptr() for i in range(n)])
        assert e.value.code == -7


def test_patch_matrix_bit_exact_yuy2(gpu, oracle, weights_tiny):
    """the IR pipeline's capture format (src/pipeline_ir.rs:27-41), 640x512"""
    rng = np.random.default_rng(9)
    w, h = 640, 512
    buf = rng.integers(0, 256, 2 * w * h, dtype=np.uint8)
    got, want, r_gpu, r_ref = _patches_case(gpu, oracle, weights_tiny, gpu.YUY2Frame(buf, w, h),
                                            oracle.Frame.yuy2(buf, w, h), (301, 203, 70, 50))
    assert np.array_equal(got, want)
    assert max(abs(a - b) for a, b in zip(r_gpu.bbox, r_ref.bbox)) <= 1


def test_patch_matrix_bit_exact_rgb8(gpu, oracle, weights_tiny):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    got, want, _, _ = _patches_case(gpu, oracle, weights_tiny, img, oracle.Frame.rgb8(img),
                                    (250, 190, 90, 60))
    assert np.array_equal(got, want)


# ---- stage (c): the network, per-stage tolerances ------------------------------------------------

def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def _assert_same_pair(got, want):
    """the first tensor of the residual stream (patch embedding + bias + positions), HIP against the oracle: both round the
    same float32 value - up to its summation order - to the 3-byte pair hi + lo8 * 2^-12 (numerical specification v3), so
    an element either agrees exactly or sits ONE quantum of the low half away (a rounding boundary in between: 2^-12
    absolute, 8.6e-5 of this tensor's maximum; until round 5, with a bf16 low half, the bar was 2e-5 of the maximum), and
    such elements are few"""
    d = np.abs(got - want)
    assert d.max() <= 2.0 ** -12 * 1.0001, d.max()
    assert (d > 0).mean() < 0.02, (d > 0).mean()


# residual stream (max over blocks), final features, head logits: relative to the tensor's maximum
# measured (round 3, spec v2): tiny 1.2e-4 / 2.3e-3 / 1.0e-3, cfg2 1.2e-3 / 4.0e-3 / 4.4e-3,
# cfg3 1.1e-3 / 3.9e-3 / 5.3e-3, cfg5 2.1e-3 / 4.1e-3 / 5.2e-3
# round 6, specification v3 (3-byte residual pair, absolute quantum 2^-12): tiny 3.4e-4 / 4.7e-3 / 1.8e-3, cfg2 1.6e-3 / 4.0e-3 /
# 4.5e-3, cfg3 1.3e-3 / 3.9e-3 / 5.1e-3, cfg5 2.7e-3 / 4.1e-3 / 6.1e-3 - only the tiny model's residual bar moved (5e-4 -> 1e-3:
# its stream is small, one quantum is 1e-4 of its maximum)
TAP_BARS = {"tiny": (1e-3, 6e-3, 3e-3), "cfg2": (3e-3, 8e-3, 1e-2), "cfg3": (3e-3, 8e-3, 1.2e-2),
            "cfg5": (5e-3, 9e-3, 1.2e-2)}


@pytest.mark.parametrize("cfg", ["tiny", "cfg2", "cfg3", "cfg5"])
def test_network_stage_taps(gpu, oracle, cfg):
    """residual stream after the patch embedding and after EVERY encoder block, final features and
    head logits against the oracle - including the 720- and 980-token shapes of the headline
    configurations (one oracle forward each: ~3 s and ~15 s)"""
    weights = gpu.weights.ensure_weights(cfg)
    sc = gpu.synth.MovingSquare(640, 480, 64, seed=2)
    buf = sc.frame_nv12(0)
    box = sc.gt_box(0)
    trk = gpu.VitTrack.new(weights)
    g = trk.as_group()
    g.enable_taps(True)
    ref = oracle.VitTrackRef(weights)
    f = gpu.NV12Frame(buf, 640, 480)
    of = oracle.Frame.nv12(buf, 640, 480)
    trk.init(f, gpu.BBox.new(*box))
    ref.init(of, box)
    r_gpu = trk.update(f)
    r_ref = ref.update(of, taps=True)
    mi = trk.model_info()
    n, d = mi.tokens_template + mi.tokens_search, mi.dim
    assert np.array_equal(g.read_tensor("patches").reshape(n, mi.kpad),
                          oracle.bf16_bits_to_f32(ref.last["patches"]))
    tok0 = g.read_tensor("tokens0").reshape(n, d)
    _assert_same_pair(tok0, ref.last["tokens0"])
    # bars = 2-3 x what was measured (tools/arbiter.py): both implementations round
    # to bf16 at the same points, what differs is float32 summation order, so a one-ulp bf16 flip here and
    # there is all there is - a kernel that makes the stream ten times worse must fail
    bar_x, bar_feat, bar_head = TAP_BARS[cfg]
    worst = max(_rel(g.read_tensor(f"layer{l}").reshape(n, d), ref.last[f"layer{l}"]) for l in range(mi.layers))
    feat = g.read_tensor("feat").reshape(mi.tokens_search, d)
    ho = g.read_tensor("head_out").reshape(mi.tokens_search, 8)
    e_head = np.abs(ho[:, :5] - ref.last["head_out"][:, :5]).max() / max(1.0, np.abs(ref.last["head_out"]).max())
    print(f"\n[taps {cfg}] residual worst layer {worst:.2e} (bar {bar_x:.0e}), feat {_rel(feat, ref.last['feat']):.2e} "
          f"(bar {bar_feat:.0e}), head logits {e_head:.2e} of max (bar {bar_head:.0e})")
    assert worst < bar_x
    assert _rel(feat, ref.last["feat"]) < bar_feat
    assert e_head < bar_head
    assert abs(r_gpu.score - r_ref.score) < 0.02
    g.enable_taps(False)


@pytest.mark.parametrize("B", [30, 8])
def test_network_stage_taps_on_the_30_stream_engine(gpu, oracle, weights_cfg3, B):
    """the same per-block comparison on the benchmark's shapes: one engine of 30 cfg3 streams runs every
    encoder GEMM on the 256x256 kernels (persistent for QKV / fc1; the row terms of the folded LayerNorms
    finalized by the last workgroup of each row panel); the first and the last stream's residual streams
    after every block, features and head logits against one oracle forward. 8 streams: the mixed case -
    proj / fc2 on the 4-wave kernel, QKV / fc1 on the 256x256 one, row terms by the finalize launch
    (a single tracker, test above: everything on the 4-wave kernel, row terms combined in the consumer)"""
    sc = gpu.synth.MovingSquare(640, 480, 64, seed=2)
    buf, box = sc.frame_nv12(0), sc.gt_box(0)
    grp = gpu.Group(weights_cfg3, n_streams=B)
    grp.enable_taps(True)
    f = gpu.NV12Frame(buf, 640, 480)
    for i in range(B):
        grp.init_host(i, f, gpu.BBox.new(*box))
    res = grp.update_host([f] * B)
    ref = oracle.VitTrackRef(weights_cfg3)
    of = oracle.Frame.nv12(buf, 640, 480)
    ref.init(of, box)
    r_ref = ref.update(of, taps=True)
    mi = grp.model_info()
    n, d = mi.tokens_template + mi.tokens_search, mi.dim
    for i in (0, B - 1):
        assert np.array_equal(grp.read_tensor("patches", i).reshape(n, mi.kpad),
                              oracle.bf16_bits_to_f32(ref.last["patches"]))
        _assert_same_pair(grp.read_tensor("tokens0", i).reshape(n, d), ref.last["tokens0"])
        bar_x, bar_feat, bar_head = TAP_BARS["cfg3"]
        for l in range(mi.layers):
            assert _rel(grp.read_tensor(f"layer{l}", i).reshape(n, d), ref.last[f"layer{l}"]) < bar_x, (i, l)
        assert _rel(grp.read_tensor("feat", i).reshape(mi.tokens_search, d), ref.last["feat"]) < bar_feat
        ho = grp.read_tensor("head_out", i).reshape(mi.tokens_search, 8)
        assert np.abs(ho[:, :5] - ref.last["head_out"][:, :5]).max() < bar_head * max(1.0, np.abs(ref.last["head_out"]).max())
        assert abs(res[i].score - r_ref.score) < 0.02 and np.abs(np.array(res[i].bbox) - np.array(r_ref.bbox)).max() <= 1


# ---- stage (d): closed-loop trajectories ------------------------------------------------------------

def _run_pair(gpu, oracle, weights, sc, frames, use_nv12=True, use_graph=True):
    w, h = sc.w, sc.h
    trk = gpu.VitTrack(weights, use_graph=use_graph)
    ref = oracle.VitTrackRef(weights)
    boxes_g, boxes_r, scores = [], [], []
    for t in range(frames):
        if use_nv12:
            buf = sc.frame_nv12(t)
            f, of = gpu.NV12Frame(buf, w, h), oracle.Frame.nv12(buf, w, h)
        else:
            img = sc.frame_rgb8(t)
            f, of = img, oracle.Frame.rgb8(img)
        if t == 0:
            # the reference host inits then updates on the SAME frame (tracker_context.rs:88-90)
            trk.init(f, gpu.BBox.new(*sc.gt_box(0)))
            ref.init(of, sc.gt_box(0))
        rg, rr = trk.update(f), ref.update(of)
        boxes_g.append(tuple(rg.bbox))
        boxes_r.append(tuple(rr.bbox))
        scores.append((rg.score, rr.score, rg.success, rr.success))
    return boxes_g, boxes_r, scores


def _assert_parity(sc, boxes_g, boxes_r, scores, min_gt_iou=0.5):
    d = np.abs(np.array(boxes_g) - np.array(boxes_r))
    ious = [iou(a, b) for a, b in zip(boxes_g, boxes_r)]
    gt_iou = [iou(a, sc.gt_box(t)) for t, a in enumerate(boxes_r)]
    assert d.max() <= 1, f"max |delta| = {d.max()} px at frame {int(d.max(axis=1).argmax())}"
    assert np.mean(ious) >= 0.99, f"mean IoU(hip, oracle) = {np.mean(ious):.4f}"
    assert all(s[2] == s[3] for s in scores), "success flags differ"
    assert max(abs(s[0] - s[1]) for s in scores) < 0.03
    # the fitted head really follows the square (otherwise the parity above would be vacuous)
    assert min(gt_iou) > min_gt_iou, f"oracle lost the target: min IoU vs GT {min(gt_iou):.3f}"


def test_trajectory_cfg1_rgb_640x480(gpu, oracle, weights_tiny):
    sc = gpu.synth.MovingSquare(640, 480, 64, seed=0)
    bg, br, s = _run_pair(gpu, oracle, weights_tiny, sc, 300, use_nv12=False)
    _assert_parity(sc, bg, br, s)


# The 300-frame trajectories of cfg2 / cfg3 and the 60-frame 4K ViT-L/14 one (cfg5) are compared with
# COMMITTED oracle fixtures in tests/test_gpu_trajectories.py (the live oracle costs 1-10 s per frame).


def test_graph_and_eager_agree(gpu, weights_tiny):
    sc = gpu.synth.MovingSquare(640, 480, 64, seed=4)
    out = []
    for use_graph in (True, False):
        trk = gpu.VitTrack(weights_tiny, use_graph=use_graph)
        boxes = []
        for t in range(20):
            f = gpu.NV12Frame(sc.frame_nv12(t), 640, 480)
            if t == 0:
                trk.init(f, gpu.BBox.new(*sc.gt_box(0)))
            r = trk.update(f)
            boxes.append((tuple(r.bbox), r.score))
        out.append(boxes)
    assert out[0] == out[1]


def test_group_streams_are_independent(gpu, weights_tiny):
    """B streams batched in one pass give exactly what B single trackers give."""
    import torch
    B, w, h = 3, 640, 480
    scs = [gpu.synth.MovingSquare(w, h, 64, seed=10 + i) for i in range(B)]
    grp = gpu.Group(weights_tiny, n_streams=B)
    singles = [gpu.VitTrack(weights_tiny) for _ in range(B)]
    for t in range(8):
        bufs = [torch.from_numpy(sc.frame_nv12(t)).cuda() for sc in scs]
        frames = [gpu.frame_nv12(b.data_ptr(), b.data_ptr() + w * h, w, h) for b in bufs]
        if t == 0:
            for i in range(B):
                grp.init_device(i, frames[i], gpu.BBox.new(*scs[i].gt_box(0)))
                singles[i].init(gpu.NV12Frame(scs[i].frame_nv12(0), w, h),
                                gpu.BBox.new(*scs[i].gt_box(0)))
        res = grp.update_device(frames)
        for i in range(B):
            r1 = singles[i].update(gpu.NV12Frame(scs[i].frame_nv12(t), w, h))
            assert res[i].bbox == r1.bbox and abs(res[i].score - r1.score) < 1e-6
    st = grp.read_state(1)
    assert st["frames_done"] == 8


def test_batched_30_streams_cfg3_large_tile_path(gpu, oracle, weights_cfg3):
    """30 streams of ViT-B/16 t192/s384 in one pass (M = 21,600 rows): every encoder GEMM runs on
    the 256x256 8-wave kernel and the attention on the LDS-DMA kernel, which the few-stream tests
    never reach. Three sample streams are checked against the CPU oracle for 3 frames (box +-1 px,
    score), and every stream must follow its own square (no cross-talk between streams)."""
    import torch
    B, w, h = 30, 1920, 1080
    scs = [gpu.synth.MovingSquare(w, h, 64, seed=100 + i) for i in range(B)]
    grp = gpu.Group(weights_cfg3, n_streams=B)
    picks = [0, 13, 29]
    refs = {i: oracle.VitTrackRef(weights_cfg3) for i in picks}
    for t in range(3):
        host = [sc.frame_nv12(t) for sc in scs]
        bufs = [torch.from_numpy(b).cuda() for b in host]
        frames = [gpu.frame_nv12(b.data_ptr(), b.data_ptr() + w * h, w, h) for b in bufs]
        if t == 0:
            for i in range(B):
                grp.init_device(i, frames[i], gpu.BBox.new(*scs[i].gt_box(0)))
            for i in picks:
                refs[i].init(oracle.Frame.nv12(host[i], w, h), scs[i].gt_box(0))
        res = grp.update_device(frames)
        for i in range(B):
            gx, gy, gw, gh = scs[i].gt_box(t)
            bx, by, bw, bh = res[i].bbox
            assert res[i].success and abs(bx - gx) <= 4 and abs(by - gy) <= 4, (t, i, res[i])
        for i in picks:
            r = refs[i].update(oracle.Frame.nv12(host[i], w, h))
            d = np.abs(np.array(res[i].bbox) - np.array(r.bbox)).max()
            assert d <= 1 and abs(res[i].score - r.score) < 0.03, (t, i, res[i], r)


def test_large_engine_keeps_the_256x256_kernels_and_matches_the_30_stream_engine(gpu, weights_cfg3):
    """vt_plan_engines hands out engines of up to 970 cfg3 streams; the 256x256 kernels address their
    operands with unsigned 32-bit byte offsets (fc2's A operand of 330 streams is 1.46 GB, beyond the
    signed range the round-2 kernels stopped at). One engine of 330 streams (M = 237,600): every stream
    is fed the same two frames, so (a) all 330 results must be identical to each other and (b) equal to
    what the 30-stream engine gives for that input up to the per-kernel tile order (+-1 px, 0.02)."""
    B, w, h = 330, 1920, 1080
    sc = gpu.synth.MovingSquare(w, h, 64, seed=411)
    big = gpu.Group(weights_cfg3, n_streams=B)
    small = gpu.Group(weights_cfg3, n_streams=30)
    f0, f1 = gpu.NV12Frame(sc.frame_nv12(0), w, h), gpu.NV12Frame(sc.frame_nv12(1), w, h)
    box = gpu.BBox.new(*sc.gt_box(0))
    for i in range(B):
        big.init_host(i, f0, box)
    for i in range(30):
        small.init_host(i, f0, box)
    for f in (f0, f1):
        rb, rs = big.update_host([f] * B), small.update_host([f] * 30)
        assert all(r.bbox == rb[0].bbox and r.score == rb[0].score and r.success for r in rb)
        assert np.abs(np.array(rb[0].bbox) - np.array(rs[0].bbox)).max() <= 1 and abs(rb[0].score - rs[0].score) < 0.02
    for i in (0, 157, 329):
        assert big.read_state(i)["frames_done"] == 2 and big.read_state(i)["success_count"] == 2
    # the per-kernel profile names the tile configuration launch_gemm() really ran: all encoder GEMMs 256x256
    import torch
    d = torch.from_numpy(sc.frame_nv12(2)).cuda()
    fr = gpu.frame_nv12(d.data_ptr(), d.data_ptr() + w * h, w, h)
    names = [k["name"] for k in big.profile_device([fr] * B, iters=1) if k["name"].startswith("gemm") and "relu" not in k["name"]]
    assert len(names) == 5 and all("256x256" in nm for nm in names), names


def test_planned_engines_run_concurrently_and_track_like_one_engine(gpu, weights_cfg3):
    """vt_plan_engines splits 33 ViT-B/16 streams into engines of 30 + 3 (no engine just past a GEMM
    round boundary). Both engines are enqueued before either is waited for, so their kernels share
    the GPU; every stream must track its square, and agree with the same stream run in ONE engine of
    33 (different tile kernels, so +-1 px and 0.02 in score rather than bit-equality)."""
    import torch
    n, w, h = 33, 1920, 1080
    sizes = gpu.plan_engines(gpu.Group(weights_cfg3, n_streams=1).model_info(), n)
    assert sizes == [30, 3] and sizes == gpu.weights.plan_engines("cfg3", n)
    scs = [gpu.synth.MovingSquare(w, h, 64, seed=300 + i) for i in range(n)]
    engines = [gpu.Group(weights_cfg3, n_streams=b) for b in sizes]
    one = gpu.Group(weights_cfg3, n_streams=n)
    off = [0, sizes[0], n]
    for t in range(4):
        bufs = [torch.from_numpy(sc.frame_nv12(t)).cuda() for sc in scs]
        frames = [gpu.frame_nv12(b.data_ptr(), b.data_ptr() + w * h, w, h) for b in bufs]
        if t == 0:
            for i in range(n):
                e = 0 if i < sizes[0] else 1
                engines[e].init_device(i - off[e], frames[i], gpu.BBox.new(*scs[i].gt_box(0)))
                one.init_device(i, frames[i], gpu.BBox.new(*scs[i].gt_box(0)))
            continue
        for e in range(2):
            engines[e].enqueue_device(frames[off[e]:off[e + 1]])
        res = engines[0].wait() + engines[1].wait()
        ref = one.update_device(frames)
        for i in range(n):
            gx, gy, _, _ = scs[i].gt_box(t)
            assert res[i].success and abs(res[i].bbox[0] - gx) <= 4 and abs(res[i].bbox[1] - gy) <= 4, (t, i, res[i])
            d = np.abs(np.array(res[i].bbox) - np.array(ref[i].bbox)).max()
            assert d <= 1 and abs(res[i].score - ref[i].score) < 0.02, (t, i, res[i], ref[i])


def test_group_host_frames_equal_device_frames(gpu, weights_tiny):
    """vt_group_update_host: B host frames of mixed pixel formats in one call (windows packed into
    one pinned arena, one H2D copy) give exactly what the same frames resident in HBM give."""
    import torch
    B, w, h = 4, 640, 480
    scs = [gpu.synth.MovingSquare(w, h, 64, seed=40 + i) for i in range(B)]
    g_host = gpu.Group(weights_tiny, n_streams=B)
    g_dev = gpu.Group(weights_tiny, n_streams=B)

    def host_frame(i, t):     # streams 0, 2: NV12; 1: RGB8; 3: YUY2 made from the same scene
        if i == 1:
            return scs[i].frame_rgb8(t)
        if i == 3:
            return gpu.YUY2Frame(scs[i].frame_yuy2(t), w, h) if hasattr(scs[i], "frame_yuy2") else \
                gpu.NV12Frame(scs[i].frame_nv12(t), w, h)
        return gpu.NV12Frame(scs[i].frame_nv12(t), w, h)

    for t in range(6):
        hf = [host_frame(i, t) for i in range(B)]
        keep, dframes = [], []
        for i, f in enumerate(hf):
            if isinstance(f, gpu.NV12Frame):
                d = torch.from_numpy(f.buf).cuda()
                dframes.append(gpu.frame_nv12(d.data_ptr(), d.data_ptr() + w * h, w, h))
            elif isinstance(f, gpu.YUY2Frame):
                d = torch.from_numpy(f.buf).cuda()
                dframes.append(gpu.CFrame(d.data_ptr(), None, w, h, 2 * w, 0, gpu.PIX_YUY2, 0, 0, 0, 0, 0))
            else:
                d = torch.from_numpy(np.ascontiguousarray(f)).cuda()
                dframes.append(gpu.frame_rgb8(d.data_ptr(), w, h))
            keep.append(d)
        if t == 0:
            for i in range(B):
                box = gpu.BBox.new(*scs[i].gt_box(0))
                g_host.init_host(i, hf[i], box)
                g_dev.init_device(i, dframes[i], box)
        rh = g_host.update_host(hf)
        rd = g_dev.update_device(dframes)
        for i in range(B):
            assert rh[i].bbox == rd[i].bbox and rh[i].success == rd[i].success
            assert abs(rh[i].score - rd[i].score) < 1e-6
            assert rh[i].success
    with pytest.raises(gpu.VtError):
        g_host.update_host(hf[:2])          # a pass needs one frame per stream


def test_dmabuf_import_roundtrip_tracks_like_the_source_buffer(gpu, weights_tiny):
    """vt_import_dmabuf: a frame that lives in a dma-buf is tracked through the mapping, without a
    host copy. The only dma-buf exporter on this machine is the GPU itself, so the test exports a
    device allocation (vt_export_dmabuf), imports the fd back and checks that tracking through the
    mapping gives exactly the results of tracking on the original pointer while a new frame is
    written into the buffer every step (so the mapping must alias the same memory). Skips if the driver refuses either step."""
    import os
    import torch
    w, h = 640, 480
    sc = gpu.synth.MovingSquare(w, h, 64, seed=21)
    # the export names a whole allocation: a range that starts inside one comes back mapped from the allocation's base (a
    # page-aligned sub-range of a cached torch segment was tried: the mapping then aliases other bytes). So the buffer must be
    # an allocation of its own - with the caching allocator's free blocks released first, a request of 32 MiB (beyond the size
    # it serves from shared segments) is one
    torch.cuda.empty_cache()
    nbytes = 32 << 20
    buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    with pytest.raises(gpu.VtError):                    # a range that starts inside an allocation is refused, not aliased
        gpu.export_dmabuf(buf.data_ptr() + 4096, 1 << 20)
    try:
        fd = gpu.export_dmabuf(buf.data_ptr(), nbytes)
    except gpu.VtError as e:
        pytest.skip(f"dma-buf export not available here: {e}")
    try:
        try:
            mapped = gpu.DmaBuf(fd, nbytes)
        except gpu.VtError as e:
            pytest.skip(f"dma-buf import refused by the driver: {e}")
        fb = w * h * 3 // 2
        t_src, t_map = gpu.VitTrack(weights_tiny), gpu.VitTrack(weights_tiny)
        for t in range(6):
            frame = torch.from_numpy(sc.frame_nv12(t)).cuda()
            buf[:fb] = frame                             # the "capture" writes into the dma-buf
            torch.cuda.synchronize()
            if t == 0:
                box = gpu.BBox.new(*sc.gt_box(0))
                t_src.init_nv12_device(buf.data_ptr(), buf.data_ptr() + w * h, w, h, w, w, box)
                t_map.init_nv12_device(mapped.ptr, mapped.ptr + w * h, w, h, w, w, box)
            a = t_src.update_nv12_device(buf.data_ptr(), buf.data_ptr() + w * h, w, h, w, w)
            b = t_map.update_nv12_device(mapped.ptr, mapped.ptr + w * h, w, h, w, w)
            assert a.success and a.bbox == b.bbox and abs(a.score - b.score) < 1e-6
        before = set(os.listdir("/proc/self/fd"))
        mapped.close()
        gone = before - set(os.listdir("/proc/self/fd"))
        assert len(gone) == 1, f"the release closes the descriptor the import duplicated (ROCm keeps it open): {gone}"
    finally:
        os.close(fd)


def test_errors_do_not_abort(gpu, weights_tiny, tmp_path):
    with pytest.raises(gpu.VtError):
        gpu.VitTrack.new(str(tmp_path / "missing.vtw"))
    bad = tmp_path / "bad.vtw"
    bad.write_bytes(b"NOTAVTWB" + b"\0" * 1000)
    with pytest.raises(gpu.VtError):
        gpu.VitTrack.new(str(bad))
    trk = gpu.VitTrack.new(weights_tiny)
    img = np.zeros((480, 640, 3), np.uint8)
    with pytest.raises(gpu.VtError) as e:
        trk.update(img)          # update before init
    assert e.value.code == -6
    with pytest.raises(gpu.VtError):
        trk.init(img, gpu.BBox.new(10, 10, 0, 5))


# ---- ingest variants, blob-from-HBM, pipelined enqueue --------------------------------------------

@pytest.mark.parametrize("fmt", ["nv12", "rgb8"])
def test_host_window_upload_equals_full_device_frame(gpu, weights_tiny, fmt):
    """Host-pointer calls upload only the search window; the result must be identical to handing
    the whole frame over in HBM, including windows cut by the frame border and a target that
    leaves the frame."""
    import torch
    w, h = 640, 480
    rng = np.random.default_rng(11)
    for box in [(288, 208, 64, 64), (2, 3, 40, 50), (600, 440, 36, 36), (300, 10, 90, 30)]:
        host, dev = gpu.VitTrack(weights_tiny), gpu.VitTrack(weights_tiny)
        for t in range(4):
            if fmt == "nv12":
                buf = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
                d = torch.from_numpy(buf).cuda()
                if t == 0:
                    host.init(gpu.NV12Frame(buf, w, h), gpu.BBox.new(*box))
                    dev.init_nv12_device(d.data_ptr(), d.data_ptr() + w * h, w, h, w, w,
                                         gpu.BBox.new(*box))
                a = host.update(gpu.NV12Frame(buf, w, h))
                b = dev.update_nv12_device(d.data_ptr(), d.data_ptr() + w * h, w, h, w, w)
            else:
                img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
                d = torch.from_numpy(img).cuda()
                if t == 0:
                    host.init(img, gpu.BBox.new(*box))
                    dev.init_rgb8_device(d.data_ptr(), w, h, 3 * w, gpu.BBox.new(*box))
                a = host.update(img)
                b = dev.update_rgb8_device(d.data_ptr(), w, h, 3 * w)
            assert a.bbox == b.bbox and a.score == b.score and a.success == b.success, (box, t)


def test_strided_host_frames(gpu, weights_tiny):
    """row strides larger than the width (GstVideoMeta-style padding) give the same result"""
    import ctypes
    w, h, pad = 640, 480, 64
    sc = gpu.synth.MovingSquare(w, h, 64, seed=2)
    rgb = sc.frame_rgb8(0)
    wide = np.zeros((h, w + pad, 3), np.uint8)
    wide[:, :w] = rgb
    a, b = gpu.VitTrack(weights_tiny), gpu.VitTrack(weights_tiny)
    a.init(rgb, gpu.BBox.new(*sc.gt_box(0)))
    ra = a.update(rgb)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L = gpu.lib()
    box = gpu.BBox.new(*sc.gt_box(0))._c()
    assert L.vt_init_rgb8(b._h, wide.ctypes.data_as(u8p), w, h, (w + pad) * 3, box) == 0
    res = gpu.CResult()
    assert L.vt_update_rgb8(b._h, wide.ctypes.data_as(u8p), w, h, (w + pad) * 3,
                            ctypes.byref(res)) == 0
    assert [res.bbox.x, res.bbox.y, res.bbox.width, res.bbox.height] == ra.bbox
    assert res.score == pytest.approx(ra.score, abs=0)


def test_create_from_device_blob(gpu, weights_tiny):
    import torch
    raw = np.fromfile(weights_tiny, dtype=np.uint8)
    blob = torch.from_numpy(raw).cuda()
    g = gpu.Group(n_streams=2, device_blob=(blob.data_ptr(), blob.numel()))
    del blob                      # the library keeps its own copy
    torch.cuda.empty_cache()
    ref = gpu.Group(weights_tiny, n_streams=2)
    w, h = 640, 480
    sc = gpu.synth.MovingSquare(w, h, 64, seed=6)
    for t in range(3):
        d = torch.from_numpy(sc.frame_nv12(t)).cuda()
        fr = [gpu.frame_nv12(d.data_ptr(), d.data_ptr() + w * h, w, h)] * 2
        if t == 0:
            for grp in (g, ref):
                for i in range(2):
                    grp.init_device(i, fr[i], gpu.BBox.new(*sc.gt_box(0)))
        ra, rb = g.update_device(fr), ref.update_device(fr)
        assert [(r.bbox, r.score) for r in ra] == [(r.bbox, r.score) for r in rb]
    assert g.model_info().weight_bytes == raw.size


def test_pipelined_enqueue_matches_sync_updates(gpu, weights_tiny):
    import torch
    w, h, n = 640, 480, 24
    sc = gpu.synth.MovingSquare(w, h, 64, seed=8)
    clip = [torch.from_numpy(sc.frame_nv12(t)).cuda() for t in range(n)]
    fr = [[gpu.frame_nv12(c.data_ptr(), c.data_ptr() + w * h, w, h)] for c in clip]
    a, b = gpu.Group(weights_tiny, n_streams=1), gpu.Group(weights_tiny, n_streams=1)
    for grp in (a, b):
        grp.init_device(0, fr[0][0], gpu.BBox.new(*sc.gt_box(0)))
    for t in range(n):            # more passes in flight than descriptor-ring slots
        a.enqueue_device(fr[t])
    ra = a.wait()[0]
    for t in range(n):
        rb = b.update_device(fr[t])[0]
    assert ra.bbox == rb.bbox and ra.score == rb.score
    sa, sb = a.read_state(0), b.read_state(0)
    assert sa["frames_done"] == sb["frames_done"] == n
    assert np.array_equal(sa["box"], sb["box"])


# ---- strided ingest against the ORACLE (not HIP against HIP) -------------------

def _strided(plane, stride, fill):
    """copy a (rows, row_bytes) uint8 plane into rows of `stride` bytes; the padding holds `fill`
    (non-zero garbage: a kernel that ignored the stride would sample it)"""
    rows, rb = plane.shape
    out = np.full((rows, stride), fill, np.uint8)
    out[:, :rb] = plane
    return out


@pytest.mark.parametrize("where", ["host", "device"])
def test_strided_nv12_rgb8_yuy2_patch_matrix_vs_oracle(gpu, oracle, weights_tiny, where):
    """GstVideoMeta-style padded rows (the reference assumes stride == width,
    src/nv12_convert.rs:47-54): NV12 with y_stride != uv_stride != width, RGB8 and YUY2 with padded
    rows, through the host-pointer and the device-pointer entry points. The patch matrix must equal
    the ORACLE's, which reads the same strided buffers (and, as a cross-check of the oracle itself,
    what it computes from the packed frame)."""
    import ctypes
    import torch
    w, h = 640, 480
    rng = np.random.default_rng(23)
    box = (301, 187, 70, 54)
    mi = gpu.VitTrack(weights_tiny).model_info()
    n, kpad = mi.tokens_template + mi.tokens_search, mi.kpad
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L = gpu.lib()

    def oracle_patches(of):
        ref = oracle.VitTrackRef(weights_tiny)
        ref.init(of, box)
        ref.update(of, taps=True)
        return oracle.bf16_bits_to_f32(ref.last["patches"])

    def run_host(init_fn, update_fn):
        trk = gpu.VitTrack(weights_tiny)
        assert init_fn(trk._h) == 0, L.vt_last_error()
        res = gpu.CResult()
        assert update_fn(trk._h, ctypes.byref(res)) == 0, L.vt_last_error()
        return trk.as_group().read_tensor("patches").reshape(n, kpad), trk

    def run_device(frame):
        grp = gpu.Group(weights_tiny, n_streams=1)
        grp.init_device(0, frame, gpu.BBox.new(*box))
        grp.update_device([frame])
        return grp.read_tensor("patches").reshape(n, kpad), grp

    cbox = gpu.BBox.new(*box)._c()
    # ---- NV12: y_stride = w + 64, uv_stride = w + 32
    yp = rng.integers(0, 256, (h, w), dtype=np.uint8)
    uvp = rng.integers(0, 256, (h // 2, w), dtype=np.uint8)
    ys, uvs = w + 64, w + 32
    yb, uvb = _strided(yp, ys, 0xAB), _strided(uvp, uvs, 0xCD)
    want = oracle_patches(oracle.Frame(1, yb.reshape(-1), uvb.reshape(-1), w, h, ys, uvs))
    assert np.array_equal(want, oracle_patches(oracle.Frame.nv12(np.concatenate([yp.reshape(-1), uvp.reshape(-1)]), w, h)))
    if where == "host":
        got, keep = run_host(lambda t: L.vt_init_nv12(t, yb.ctypes.data_as(u8p), uvb.ctypes.data_as(u8p), w, h, ys, uvs, cbox),
                             lambda t, r: L.vt_update_nv12(t, yb.ctypes.data_as(u8p), uvb.ctypes.data_as(u8p), w, h, ys, uvs, r))
    else:
        dy, duv = torch.from_numpy(yb).cuda(), torch.from_numpy(uvb).cuda()
        got, keep = run_device(gpu.frame_nv12(dy.data_ptr(), duv.data_ptr(), w, h, ys, uvs))
    assert np.array_equal(got, want), "NV12 strided"
    # ---- RGB8: stride = 3 w + 21 (not even a multiple of 3)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rs = 3 * w + 21
    rb = _strided(img.reshape(h, 3 * w), rs, 0x5A)
    want = oracle_patches(oracle.Frame(0, rb.reshape(-1), None, w, h, rs, 0))
    assert np.array_equal(want, oracle_patches(oracle.Frame.rgb8(img)))
    if where == "host":
        got, keep = run_host(lambda t: L.vt_init_rgb8(t, rb.ctypes.data_as(u8p), w, h, rs, cbox),
                             lambda t, r: L.vt_update_rgb8(t, rb.ctypes.data_as(u8p), w, h, rs, r))
    else:
        d = torch.from_numpy(rb).cuda()
        got, keep = run_device(gpu.frame_rgb8(d.data_ptr(), w, h, rs))
    assert np.array_equal(got, want), "RGB8 strided"
    # ---- YUY2: stride = 2 w + 48
    yuy = rng.integers(0, 256, (h, 2 * w), dtype=np.uint8)
    qs = 2 * w + 48
    qb = _strided(yuy, qs, 0x77)
    want = oracle_patches(oracle.Frame(2, qb.reshape(-1), None, w, h, qs, 0))
    assert np.array_equal(want, oracle_patches(oracle.Frame.yuy2(yuy.reshape(-1), w, h)))
    if where == "host":
        got, keep = run_host(lambda t: L.vt_init_yuy2(t, qb.ctypes.data_as(u8p), w, h, qs, cbox),
                             lambda t, r: L.vt_update_yuy2(t, qb.ctypes.data_as(u8p), w, h, qs, r))
    else:
        d = torch.from_numpy(qb).cuda()
        got, keep = run_device(gpu.CFrame(d.data_ptr(), None, w, h, qs, 0, gpu.PIX_YUY2, 0, 0, 0, 0, 0))
    assert np.array_equal(got, want), "YUY2 strided"


# ---- upload of pass t+1 overlapped with the compute of pass t ------------------

@pytest.mark.parametrize("margin_pct,expect_redo", [(0, False), (-1, True)])
def test_pipelined_host_passes_equal_synchronous_ones(gpu, weights_tiny, margin_pct, expect_redo):
    """vt_group_enqueue_host / vt_group_wait_next (speculative windows, copy stream, two arenas) give
    exactly the results of the synchronous vt_group_update_host on the same frames - with the default
    enlargement (no redo on this clip) and with the enlargement switched off, where every moving
    target leaves its speculative window and the snapshot/redo path runs."""
    B, w, h, n = 3, 640, 480, 24
    scs = [gpu.synth.MovingSquare(w, h, 64, seed=60 + i) for i in range(B)]
    pipe = gpu.Group(weights_tiny, n_streams=B, host_window_margin_pct=margin_pct)
    sync = gpu.Group(weights_tiny, n_streams=B)
    frames = [[gpu.NV12Frame(sc.frame_nv12(t), w, h) if i != 1 else sc.frame_rgb8(t)
               for i, sc in enumerate(scs)] for t in range(n)]
    for i in range(B):
        box = gpu.BBox.new(*scs[i].gt_box(0))
        pipe.init_host(i, frames[0][i], box)
        sync.init_host(i, frames[0][i], box)
    want = [sync.update_host(frames[t]) for t in range(n)]
    got = []
    pipe.enqueue_host(frames[0])
    for t in range(1, n):
        pipe.enqueue_host(frames[t])          # upload of t overlaps the pass of t-1
        got.append(pipe.wait_next())          # results of t-1
    got.append(pipe.wait_next())
    for t in range(n):
        for i in range(B):
            assert got[t][i].bbox == want[t][i].bbox and got[t][i].score == want[t][i].score, (t, i)
            assert got[t][i].success
    assert (pipe.host_redos() > 0) == expect_redo, pipe.host_redos()
    with pytest.raises(gpu.VtError):
        pipe.wait_next()                      # nothing outstanding
    pipe.enqueue_host(frames[0]); pipe.enqueue_host(frames[1])
    with pytest.raises(gpu.VtError):
        pipe.enqueue_host(frames[2])          # two passes outstanding
    # an outstanding pass owns the stream states (its redo path rewinds to the host's copy of them):
    # everything that would advance or overwrite them behind it is refused, and changes nothing
    before = [pipe.read_state(i) for i in range(B)]
    for call in (lambda: pipe.update_host(frames[2]), lambda: pipe.wait(),
                 lambda: pipe.init_host(0, frames[0][0], gpu.BBox.new(*scs[0].gt_box(0))),
                 lambda: pipe.set_state_box(0, [10.0, 10.0, 50.0, 50.0])):
        with pytest.raises(gpu.VtError) as ei:
            call()
        assert ei.value.code == -1 and "vt_group_wait_next" in str(ei.value)      # VT_ERR_INVALID_ARG
    pipe.wait_next(); pipe.wait_next()
    after = [pipe.read_state(i) for i in range(B)]
    for i in range(B):
        assert after[i]["frames_done"] == before[i]["frames_done"] or after[i]["frames_done"] == n + 2
    # the synchronous entry point still works afterwards and continues the same state chain
    assert len(pipe.update_host(frames[2])) == B


def test_zero_copy_host_mapping_tracks_like_device_frames(gpu, oracle, weights_tiny):
    """vt_host_register: frames stay in (page-locked, device-mapped) host memory and the pixel kernel reads
    what it samples over PCIe. Same results, bit for bit, as the same frames resident in HBM - single
    tracker and a group; the patch matrix of the last update is bit-exact with the oracle's."""
    import torch
    w, h, n = 640, 480, 10
    sc = gpu.synth.MovingSquare(w, h, 64, seed=17)
    clip = np.stack([sc.frame_nv12(t) for t in range(n)])
    fb = clip.shape[1]
    dclip = torch.from_numpy(clip).cuda()
    hm = gpu.HostMapping(clip)
    assert hm.d_ptr
    try:
        fr_d = lambda t: gpu.frame_nv12(dclip.data_ptr() + t * fb, dclip.data_ptr() + t * fb + w * h, w, h)
        fr_z = lambda t: gpu.frame_nv12(hm.d_ptr + t * fb, hm.d_ptr + t * fb + w * h, w, h)
        ga, gb = gpu.Group(weights_tiny, n_streams=2), gpu.Group(weights_tiny, n_streams=2)
        for i in range(2):
            ga.init_device(i, fr_d(0), gpu.BBox.new(*sc.gt_box(0)))
            gb.init_device(i, fr_z(0), gpu.BBox.new(*sc.gt_box(0)))
        for t in range(n):
            ra, rb = ga.update_device([fr_d(t)] * 2), gb.update_device([fr_z(t)] * 2)
            assert [(r.bbox, r.score, r.success) for r in ra] == [(r.bbox, r.score, r.success) for r in rb], t
        mi = ga.model_info()
        pa = ga.read_tensor("patches").reshape(-1, mi.kpad)
        pb = gb.read_tensor("patches").reshape(-1, mi.kpad)
        assert np.array_equal(pa, pb)
        trk = gpu.VitTrack.new(weights_tiny)
        trk.init_nv12_device(hm.d_ptr, hm.d_ptr + w * h, w, h, w, w, gpu.BBox.new(*sc.gt_box(0)))
        r = trk.update_nv12_device(hm.d_ptr + fb, hm.d_ptr + fb + w * h, w, h, w, w)
        assert r.success
    finally:
        del ga, gb
        hm.close()
    with pytest.raises(gpu.VtError):
        gpu.HostMapping(np.zeros(0, np.uint8))


def test_host_pointer_calls_take_the_zero_copy_route_for_registered_frames(gpu, weights_tiny):
    """a host that registered its capture pool keeps calling update(host pointer): frames inside a registered
    range skip window packing and the staging copy (vt_host_register). Results are those of the same calls on
    unregistered memory, bit for bit - single tracker (the automatic route), group synchronous and group pipelined
    (opted in with vt_config.host_zero_copy = 1; by default a batched engine keeps packing windows, which is faster
    for it: vittrack_hip.h at vt_host_register) - and after vt_host_unregister the same buffers go through the
    staging path again."""
    w, h, n, B = 640, 480, 8, 2
    sc = gpu.synth.MovingSquare(w, h, 64, seed=23)
    clip = np.stack([sc.frame_nv12(t) for t in range(n)])          # to be registered
    plain = clip.copy()                                             # never registered

    def run_single(buf):
        trk = gpu.VitTrack.new(weights_tiny)
        trk.init(gpu.NV12Frame(buf[0], w, h), gpu.BBox.new(*sc.gt_box(0)))
        return [(r.bbox, r.score, r.success) for r in (trk.update(gpu.NV12Frame(buf[t], w, h)) for t in range(n))]

    def run_group(buf, pipelined, zc=0):
        g = gpu.Group(weights_tiny, n_streams=B, host_zero_copy=zc)
        for i in range(B):
            g.init_host(i, gpu.NV12Frame(buf[0], w, h), gpu.BBox.new(*sc.gt_box(0)))
        out = []
        if pipelined:
            g.enqueue_host([gpu.NV12Frame(buf[1], w, h)] * B)
            for t in range(2, n):
                g.enqueue_host([gpu.NV12Frame(buf[t], w, h)] * B)
                out.append([(r.bbox, r.score) for r in g.wait_next()])
            out.append([(r.bbox, r.score) for r in g.wait_next()])
            assert g.host_redos() == 0 or zc <= 0          # whole mapped frames cannot miss their window
        else:
            for t in range(1, n):
                out.append([(r.bbox, r.score) for r in g.update_host([gpu.NV12Frame(buf[t], w, h)] * B)])
        return out

    want = (run_single(plain), run_group(plain, False), run_group(plain, True))
    hm = gpu.HostMapping(clip)
    try:
        got = (run_single(clip), run_group(clip, False, 1), run_group(clip, True, 1))
        dflt = (run_group(clip, False), run_group(clip, True))          # batched engines: packed windows by default
        never = run_group(clip, True, -1)
    finally:
        hm.close()
    assert got == want
    assert dflt == want[1:] and never == want[2]
    assert run_single(clip) == want[0]                      # unregistered again: the staging path
    # a frame that only STARTS inside a registered range (its tail lies in unregistered memory) must be staged,
    # not read through the mapping: register all but the last 4 KiB of a buffer that holds one frame at its end
    fb = clip.shape[1]
    big = np.zeros(fb + 8192, np.uint8)
    big[8192:] = clip[0]
    part = gpu.HostMapping(big[: fb + 8192 - 4096])
    try:
        trk = gpu.VitTrack.new(weights_tiny)
        f = gpu.NV12Frame(big[8192:], w, h)
        trk.init(f, gpu.BBox.new(*sc.gt_box(0)))
        r = trk.update(f)
        assert (r.bbox, r.score, r.success) == want[0][0]
    finally:
        part.close()


def test_zero_copy_sees_a_registered_buffer_rewritten_in_place(gpu, weights_tiny):
    """a capture pool recycles its buffers: the CPU rewrites the SAME registered buffer with frame t before every
    update. The pixel kernel must see the new bytes each time (hipHostRegister's mapping is coherent: nothing of
    an earlier frame may be served from a GPU cache) - single tracker (automatic zero-copy route), a group
    synchronous and a group pipelined with two alternating registered buffers (host_zero_copy = 1; a pipelined
    caller may not touch a buffer again before wait_next has returned its pass). Bit-identical to the staged path
    on unregistered memory."""
    w, h, n, B = 640, 480, 10, 2
    sc = gpu.synth.MovingSquare(w, h, 64, seed=29)
    clip = np.stack([sc.frame_nv12(t) for t in range(n)])
    fb = clip.shape[1]

    def run_single(frame_of):
        trk = gpu.VitTrack.new(weights_tiny)
        trk.init(frame_of(0), gpu.BBox.new(*sc.gt_box(0)))
        return [(r.bbox, r.score, r.success) for r in (trk.update(frame_of(t)) for t in range(n))]

    def run_sync(frame_of, zc):
        g = gpu.Group(weights_tiny, n_streams=B, host_zero_copy=zc)
        f0 = frame_of(0)
        for i in range(B):
            g.init_host(i, f0, gpu.BBox.new(*sc.gt_box(0)))
        return [[(r.bbox, r.score) for r in g.update_host([frame_of(t)] * B)] for t in range(1, n)]

    def run_pipe(frame_of, zc):
        g = gpu.Group(weights_tiny, n_streams=B, host_zero_copy=zc)
        f0 = frame_of(0)
        for i in range(B):
            g.init_host(i, f0, gpu.BBox.new(*sc.gt_box(0)))
        out = []
        g.enqueue_host([frame_of(1)] * B)
        for t in range(2, n):
            g.enqueue_host([frame_of(t)] * B)      # buffer t & 1: the pass that read it (t - 2) has been waited for
            out.append([(r.bbox, r.score) for r in g.wait_next()])
        out.append([(r.bbox, r.score) for r in g.wait_next()])
        return out

    plain = lambda t: gpu.NV12Frame(clip[t], w, h)
    want = (run_single(plain), run_sync(plain, -1), run_pipe(plain, -1))
    pool = np.zeros((2, fb), np.uint8)                       # the "capture pool": two recycled buffers
    hm = gpu.HostMapping(pool)

    def recycled(t):
        pool[t & 1][:] = clip[t]                             # the CPU writes frame t over what the buffer held
        return gpu.NV12Frame(pool[t & 1], w, h)

    try:
        got = (run_single(recycled), run_sync(recycled, 1), run_pipe(recycled, 1))
    finally:
        hm.close()
    assert got[0] == want[0], "single tracker: stale pixels from a rewritten registered buffer"
    assert got[1] == want[1], "synchronous group"
    assert got[2] == want[2], "pipelined group"


@pytest.mark.parametrize("cfg,B", [("tiny", 1), ("tiny", 5), ("cfg3", 1), ("cfg3", 7), ("cfg2", 3)])
def test_head_band_kernel_equals_the_separate_launches(gpu, cfg, B):
    """the head on the band kernel (k_head.hip: 4 launches, logits + decode fused behind the last 3x3 layer, the last
    band of a stream to arrive decodes it) against the same engine with the head as implicit GEMMs + head_out +
    decode launches (vt_group_set_tuning "head_band" 0): the last feature map bit-identical, the logits equal up to
    the f32 summation order of the 5-logit layer, identical boxes and success flags over a short closed loop; graph
    replay and eager."""
    name = {"tiny": "tiny", "cfg3": "vitb16_t192_s384", "cfg2": "vitb16_t128_s256"}[cfg]
    weights = gpu.weights.ensure_weights(name)
    w, h = (640, 480) if cfg == "tiny" else (1920, 1080)
    sc = gpu.synth.MovingSquare(w, h, 64, seed=41)
    for use_graph in (True, False):
        ga = gpu.Group(weights, n_streams=B, use_graph=use_graph)
        gb = gpu.Group(weights, n_streams=B, use_graph=use_graph)
        gb.set_tuning("head_band", 0)
        f0 = gpu.NV12Frame(sc.frame_nv12(0), w, h)
        for i in range(B):
            ga.init_host(i, f0, gpu.BBox.new(*sc.gt_box(0)))
            gb.init_host(i, f0, gpu.BBox.new(*sc.gt_box(0)))
        for t in range(6):
            f = gpu.NV12Frame(sc.frame_nv12(t), w, h)
            ra, rb = ga.update_host([f] * B), gb.update_host([f] * B)
            assert [(r.bbox, r.success) for r in ra] == [(r.bbox, r.success) for r in rb], t
            assert max(abs(x.score - y.score) for x, y in zip(ra, rb)) < 1e-5
            for i in (0, B - 1):
                assert np.array_equal(ga.read_tensor("head_t3", i), gb.read_tensor("head_t3", i))
                ha, hb = ga.read_tensor("head_out", i), gb.read_tensor("head_out", i)
                assert np.abs(ha - hb).max() <= 1e-5 * max(1.0, np.abs(hb).max())
        sa, sb = ga.read_state(B - 1), gb.read_state(B - 1)
        assert sa["frames_done"] == sb["frames_done"] == 6 and sa["last_idx"] == sb["last_idx"]
        del ga, gb


@pytest.mark.parametrize("cfg,B", [("cfg3", 1), ("cfg3", 30), ("cfg2", 5), ("cfg5", 2)])
def test_final_layernorm_inside_the_heads_first_kernel_changes_no_value(gpu, cfg, B):
    """the default pass (the band kernel of the head's 1x1 layer normalises its rows itself: no LayerNorm launch, "feat"
    never written by the pass) against the same engine with the LayerNorm kernel in front (vt_group_set_tuning
    "head_band" 1): every head tensor, the logits, scores, boxes and stream states bit-identical over a short closed
    loop, graph replay and eager; "feat" read from the default engine (recomputed on demand from the residual stream
    the pass left) equals the tensor the other engine's pass wrote."""
    name = {"cfg3": "vitb16_t192_s384", "cfg2": "vitb16_t128_s256", "cfg5": "vitl14_t196_s392"}[cfg]
    weights = gpu.weights.ensure_weights(name)
    w, h = 1920, 1080
    sc = gpu.synth.MovingSquare(w, h, 64, seed=43)
    for use_graph in (True, False):
        ga = gpu.Group(weights, n_streams=B, use_graph=use_graph)
        gb = gpu.Group(weights, n_streams=B, use_graph=use_graph)
        gb.set_tuning("head_band", 1)
        f0 = gpu.NV12Frame(sc.frame_nv12(0), w, h)
        for i in range(B):
            ga.init_host(i, f0, gpu.BBox.new(*sc.gt_box(0)))
            gb.init_host(i, f0, gpu.BBox.new(*sc.gt_box(0)))
        for t in range(4):
            f = gpu.NV12Frame(sc.frame_nv12(t), w, h)
            ra, rb = ga.update_host([f] * B), gb.update_host([f] * B)
            assert [(r.bbox, r.success, r.score) for r in ra] == [(r.bbox, r.success, r.score) for r in rb], t
            for i in (0, B - 1):
                for tensor in ("head_t3", "head_out", "feat"):
                    assert np.array_equal(ga.read_tensor(tensor, i), gb.read_tensor(tensor, i)), (t, i, tensor)
                assert np.array_equal(ga.read_tensor("state", i).view(np.uint32), gb.read_tensor("state", i).view(np.uint32)), (t, i)
        del ga, gb
