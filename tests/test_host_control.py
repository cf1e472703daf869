"""The C++ host mirror (libvittrack_host.so) against the Python restatement of
/root/reference/src/tracker_context.rs, selection_state.rs and timing_stats.rs, plus known-answer
cases derived from the source. CPU only: the tracker behind the context is scripted."""
import random

import numpy as np
import pytest

from oracle import tracker_context_ref as ref


@pytest.fixture(scope="module")
def hostlib(vt):
    from harness import hostlib as h
    h.lib()
    return h


class Scripted:
    """update() replays (success, score, bbox) tuples; an Exception entry is the Err arm"""

    def __init__(self, script):
        self.script, self.i, self.inits = list(script), 0, []

    def init(self, frame, bbox):
        self.inits.append(tuple(bbox))

    def update(self, frame):
        item = self.script[min(self.i, len(self.script) - 1)]
        self.i += 1
        if isinstance(item, Exception):
            raise item
        return item


FRAME = np.zeros((48, 64, 3), np.uint8)


def _pair(hostlib, script, w=640, h=480):
    a = Scripted(script)
    b = Scripted(script)
    return hostlib.TrackerContext(width=w, height=h, tracker=a), ref.TrackerContext(b, w, h), a, b


def _snapshot_c(c):
    i = c.info()
    return (i.state_kind, i.lost_frames, bool(i.has_bbox),
            (i.current_bbox.x, i.current_bbox.y, i.current_bbox.width, i.current_bbox.height)
            if i.has_bbox else None, round(i.current_score, 6), bool(i.pending_confirm),
            i.cursor_x, i.cursor_y, i.start_x, i.start_y, i.selection_phase, c.state_name())


def _snapshot_r(r):
    return (r.state, r.lost_frames, r.current_bbox is not None, r.current_bbox,
            round(float(np.float32(r.current_score)), 6), r.pending_confirm,
            r.selection.cursor_x, r.selection.cursor_y, r.selection.start_x, r.selection.start_y,
            r.selection.phase, r.state_name())


def _start_tracking(ctx, rctx):
    for c in (ctx, rctx):
        c.handle_command(ref.CONFIRM)
        c.process_frame(FRAME)                      # first confirm: start corner
        for _ in range(7):
            c.handle_command(ref.MOVE_RIGHT, False)
        for _ in range(2):
            c.handle_command(ref.MOVE_DOWN, True)
        c.handle_command(ref.CONFIRM)


def test_init_then_update_on_same_frame_and_bbox_from_selection(hostlib):
    ctx, rctx, a, b = _pair(hostlib, [(True, 0.9, (330, 250, 60, 90))])
    _start_tracking(ctx, rctx)
    assert ctx.process_frame(FRAME) == rctx.process_frame(FRAME) == (330, 250, 60, 90)
    # selection_state.rs:39-45: start (320,240), cursor (390,340) -> x,y = min, w,h = |diff|
    assert a.inits == b.inits == [(320, 240, 70, 100)]
    assert a.i == b.i == 1                           # exactly one update after the init (:88-90)
    assert ctx.state_name() == rctx.state_name() == "TRACKING"
    assert _snapshot_c(ctx) == _snapshot_r(rctx)


def test_score_gate_is_strict(hostlib):
    # tracker_context.rs:93,122: `score > 0.25`; exactly 0.25 is rejected
    for score, ok in [(0.25, False), (0.2500001, True), (0.9, True), (0.0, False)]:
        ctx, rctx, _, _ = _pair(hostlib, [(True, score, (1, 2, 30, 40))])
        _start_tracking(ctx, rctx)
        got, want = ctx.process_frame(FRAME), rctx.process_frame(FRAME)
        assert (got is not None) == (want is not None) == ok, score
        assert ctx.state_name() == rctx.state_name() == ("TRACKING" if ok else "SELECT START")
    # success flag false also rejects
    ctx, rctx, _, _ = _pair(hostlib, [(False, 0.99, (1, 2, 30, 40))])
    _start_tracking(ctx, rctx)
    assert ctx.process_frame(FRAME) is None and rctx.process_frame(FRAME) is None


def test_lost_counter_resets_on_the_62nd_lost_frame(hostlib):
    # :142-153: Lost{0} is entered on the failing frame; each later frame increments while
    # frames <= 60; the frame that SEES frames == 61 resets. So 61 frames stay LOST, the 62nd resets.
    script = [(True, 0.9, (5, 5, 30, 30)), (True, 0.1, (5, 5, 30, 30))]
    ctx, rctx, a, b = _pair(hostlib, script)
    _start_tracking(ctx, rctx)
    for c in (ctx, rctx):
        assert c.process_frame(FRAME) is not None
        assert c.process_frame(FRAME) is None and c.state_name() == "LOST"
    for k in range(61):
        for c in (ctx, rctx):
            assert c.process_frame(FRAME) is None
            assert c.state_name() == "LOST", k
        assert _snapshot_c(ctx) == _snapshot_r(rctx)
    assert ctx.info().lost_frames == rctx.lost_frames == 61
    for c in (ctx, rctx):
        c.process_frame(FRAME)
        assert c.state_name() == "SELECT START"
    assert a.i == b.i == 2                           # the tracker is not called while LOST (:142)
    assert _snapshot_c(ctx) == _snapshot_r(rctx)


def test_update_error_paths(hostlib):
    # Err during init-update: selection reset, still selecting (:105-109)
    ctx, rctx, _, _ = _pair(hostlib, [RuntimeError("boom")])
    _start_tracking(ctx, rctx)
    assert ctx.process_frame(FRAME) is None and rctx.process_frame(FRAME) is None
    assert ctx.state_name() == rctx.state_name() == "SELECT START"
    # Err while tracking: Lost{0}, current_score untouched (:134-138)
    ctx, rctx, _, _ = _pair(hostlib, [(True, 0.8, (5, 5, 30, 30)), RuntimeError("boom")])
    _start_tracking(ctx, rctx)
    for c in (ctx, rctx):
        c.process_frame(FRAME)
        assert c.process_frame(FRAME) is None and c.state_name() == "LOST"
    assert _snapshot_c(ctx) == _snapshot_r(rctx)
    assert abs(ctx.info().current_score - 0.8) < 1e-6


def test_cancel_and_cursor_clamp_and_min_bbox(hostlib):
    ctx, rctx, _, _ = _pair(hostlib, [(True, 0.9, (5, 5, 30, 30))], w=100, h=80)
    for c in (ctx, rctx):
        for _ in range(5):
            c.handle_command(ref.MOVE_LEFT, True)      # 50 - 250 -> clamp 0
            c.handle_command(ref.MOVE_DOWN, True)      # 40 + 250 -> clamp 79
    assert (ctx.info().cursor_x, ctx.info().cursor_y) == (0, 79)
    assert _snapshot_c(ctx) == _snapshot_r(rctx)
    # bbox sides are clamped to >= 20 (selection_state.rs:42-43)
    assert hostlib.selection_bbox(100, 100, 105, 300) == (100, 100, 20, 200)
    assert hostlib.selection_bbox(50, 60, 40, 55) == (40, 55, 20, 20)
    s = ref.SelectionState(0, 0)
    s.start_x, s.start_y, s.cursor_x, s.cursor_y = 50, 60, 40, 55
    assert s.get_bbox() == (40, 55, 20, 20)
    for c in (ctx, rctx):
        c.handle_command(ref.CANCEL)
    assert _snapshot_c(ctx) == _snapshot_r(rctx)
    assert ctx.state_name() == "SELECT START"


def test_random_command_and_score_traces_agree(hostlib):
    rng = random.Random(1234)
    for trial in range(25):
        script = []
        for _ in range(400):
            r = rng.random()
            if r < 0.03:
                script.append(RuntimeError("err"))
            else:
                script.append((rng.random() < 0.95, rng.choice([0.1, 0.25, 0.26, 0.6, 0.9]),
                               (rng.randrange(600), rng.randrange(440), 20 + rng.randrange(80),
                                20 + rng.randrange(80))))
        ctx, rctx, a, b = _pair(hostlib, script)
        for step in range(500):
            for _ in range(rng.randrange(3)):
                cmd = rng.choice([0, 1, 2, 3, 4, 4, 5 if rng.random() < 0.1 else 4, 6])
                fast = rng.random() < 0.5
                ctx.handle_command(cmd, fast)
                rctx.handle_command(cmd, fast)
            assert ctx.process_frame(FRAME) == rctx.process_frame(FRAME), (trial, step)
            assert _snapshot_c(ctx) == _snapshot_r(rctx), (trial, step)
        assert a.inits == b.inits and a.i == b.i


def test_timing_stats(hostlib):
    t, r = hostlib.TimingStats(), ref.TimingStats()
    assert t.fps() == r.fps() == 0.0 and t.avg_conv_ms() == 0.0 and t.avg_track_ms() == 0.0
    for x in (t, r):
        x.add_interval(16667)
        x.add_times(1500, 900)
    assert t.fps() == pytest.approx(1e6 / 16667) and t.fps() == r.fps()
    assert t.avg_conv_ms() == r.avg_conv_ms() == 1.5 and t.avg_track_ms() == r.avg_track_ms() == 0.9
    # ring of 120 samples (timing_stats.rs:19-21): old samples fall out
    for i in range(300):
        for x in (t, r):
            x.add_interval(1000 + i)
            x.add_times(i, 2 * i)
    want = 1e6 / (sum(1000 + i for i in range(180, 300)) / 120)
    assert t.fps() == pytest.approx(want) and r.fps() == pytest.approx(want)
    assert t.avg_conv_ms() == pytest.approx(r.avg_conv_ms())
    assert t.avg_track_ms() == pytest.approx(2 * t.avg_conv_ms())
    z = hostlib.TimingStats()
    z.add_interval(0)
    assert z.fps() == 0.0                              # avg == 0 -> 0.0 (:41-45)


def test_context_new_without_gpu_fails_like_the_reference(hostlib, vt, weights_tiny):
    # TrackerContext::new propagates the VitTrack::new error ("Failed: ...", :21)
    if vt.device_count() > 0:
        return
    with pytest.raises(vt.VtError) as e:
        hostlib.TrackerContext.new(weights_tiny, 640, 480)
    assert "Failed:" in str(e.value)
