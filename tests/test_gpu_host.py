"""The reference's per-frame call sequence (probe closure, /root/reference/src/pipeline.rs:67-122)
replayed through the C++ host mirror on top of the HIP library, against the Python restatement of
the control layer on top of the oracle tracker: same commands, same frames, same states and boxes."""
import os

import numpy as np
import pytest

from oracle import tracker_context_ref as ref

pytestmark = pytest.mark.gpu


class OracleTracker:
    def __init__(self, oracle, weights, w, h, nv12):
        self.t, self.o, self.w, self.h, self.nv12 = oracle.VitTrackRef(weights), oracle, w, h, nv12

    def _f(self, frame):
        return self.o.Frame.nv12(frame, self.w, self.h) if self.nv12 else self.o.Frame.rgb8(frame)

    def init(self, frame, bbox):
        self.t.init(self._f(frame), bbox)

    def update(self, frame):
        r = self.t.update(self._f(frame))
        return r.success, r.score, r.bbox


@pytest.mark.parametrize("nv12", [True, False])
def test_probe_sequence_select_track(gpu, oracle, weights_tiny, nv12):
    from harness import hostlib
    w, h = 640, 480
    sc = gpu.synth.MovingSquare(w, h, 64, seed=5)
    ctx = hostlib.TrackerContext.new(weights_tiny, w, h)
    rctx = ref.TrackerContext(OracleTracker(oracle, weights_tiny, w, h, nv12), w, h)
    x, y, bw, bh = sc.gt_box(0)

    def both(cmd, fast=False):
        ctx.handle_command(cmd, fast)
        rctx.handle_command(cmd, fast)

    def frame(t):
        if nv12:
            buf = sc.frame_nv12(t)
            return ("nv12", buf, w, h), buf
        img = sc.frame_rgb8(t)
        return img, img

    # steer the cursor to the square's top-left corner (steps of 10), confirm, bottom-right, confirm
    def steer(tx, ty):
        while abs(rctx.selection.cursor_x - tx) >= 10:
            both(ref.MOVE_RIGHT if rctx.selection.cursor_x < tx else ref.MOVE_LEFT)
        while abs(rctx.selection.cursor_y - ty) >= 10:
            both(ref.MOVE_DOWN if rctx.selection.cursor_y < ty else ref.MOVE_UP)

    steer(x, y)
    both(ref.CONFIRM)
    f, rf = frame(0)
    assert ctx.process_frame(f) is None and rctx.process_frame(rf) is None
    assert ctx.state_name() == rctx.state_name() == "SELECT END"
    steer(x + bw, y + bh)
    both(ref.CONFIRM)
    got, want = ctx.process_frame(f), rctx.process_frame(rf)      # init + update on frame 0
    assert ctx.state_name() == rctx.state_name() == "TRACKING"
    assert max(abs(a - b) for a, b in zip(got, want)) <= 1
    for t in range(1, 40):
        f, rf = frame(t)
        got, want = ctx.process_frame(f), rctx.process_frame(rf)
        assert (got is None) == (want is None)
        assert max(abs(a - b) for a, b in zip(got, want)) <= 1, t
        assert ctx.state_name() == rctx.state_name() == "TRACKING"
    assert abs(ctx.info().current_score - rctx.current_score) < 0.03
    both(ref.CANCEL)
    assert ctx.state_name() == rctx.state_name() == "SELECT START"


def test_host_nv12_full_to_rgb_through_host_lib(gpu, oracle):
    from harness import hostlib
    import ctypes
    w, h = 320, 240
    sc = gpu.synth.MovingSquare(w, h, 48, seed=9)
    buf = np.ascontiguousarray(sc.frame_nv12(1))
    out = np.empty((h, w, 3), np.uint8)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    rc = hostlib.lib().vth_nv12_full_to_rgb(0, buf.ctypes.data_as(u8p), buf.size, w, h,
                                            out.ctypes.data_as(u8p))
    assert rc == 0
    ref_rgb, _ = oracle.nv12_to_rgb8(buf, w, h, 2)
    assert np.array_equal(out, ref_rgb)


def test_rccl_weight_broadcast_without_python_collectives(gpu, weights_tiny):
    """vt_rccl_unique_id + vt_broadcast_weights_rccl (direct ncclBroadcast through a dlopen'ed
    librccl - the path a non-Python host takes) at world size 1 on this
    one-GPU box: the blob that arrives in HBM must be the file, and a group built from it must track
    exactly like one built from the file."""
    import torch
    uid = gpu.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    ptr, nbytes = gpu.broadcast_weights_rccl(uid, 1, 0, 0, weights_tiny)
    try:
        assert nbytes == os.path.getsize(weights_tiny)
        g = gpu.Group(n_streams=1, device_blob=(ptr, nbytes))    # validates every tensor of the blob
        assert g.model_info().weight_bytes == nbytes
    finally:
        gpu.free_device_blob(ptr)
    ref = gpu.Group(weights_tiny, n_streams=1)
    w, h_ = 640, 480
    sc = gpu.synth.MovingSquare(w, h_, 64, seed=12)
    for t in range(3):
        d = torch.from_numpy(sc.frame_nv12(t)).cuda()
        fr = [gpu.frame_nv12(d.data_ptr(), d.data_ptr() + w * h_, w, h_)]
        if t == 0:
            for grp in (g, ref):
                grp.init_device(0, fr[0], gpu.BBox.new(*sc.gt_box(0)))
        ra, rb = g.update_device(fr)[0], ref.update_device(fr)[0]
        assert ra.bbox == rb.bbox and ra.score == rb.score


def test_plain_c_client_tracks_like_the_python_binding(gpu, weights_tiny, tmp_path):
    """the reference's call sequence (new -> init -> update ...) driven from a C99 program that knows
    only include/vittrack_hip.h and dlopen: same results as the ctypes binding, to the last bit"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "harness", "c_client")
    assert os.path.exists(exe), "harness/c_client missing: python __graft_entry__.py"
    w, h, n = 640, 480, 12
    sc = gpu.synth.MovingSquare(w, h, 64, seed=5)
    frames = [sc.frame_nv12(t) for t in range(n)]
    clip = tmp_path / "clip.nv12"
    clip.write_bytes(b"".join(f.tobytes() for f in frames))
    box = sc.gt_box(0)
    out = subprocess.run([exe, "run", gpu.LIB_PATH, weights_tiny, str(clip), str(w), str(h), str(n)] +
                         [str(int(v)) for v in box], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    trk = gpu.VitTrack.new(weights_tiny)
    trk.init(gpu.NV12Frame(frames[0], w, h), gpu.BBox.new(*box))
    lines = out.stdout.strip().splitlines()
    assert len(lines) == n
    for t, line in enumerate(lines):
        r = trk.update(gpu.NV12Frame(frames[t], w, h))
        tt, succ, score, x, y, bw, bh = line.split()
        assert int(tt) == t and int(succ) == int(r.success) and (int(x), int(y), int(bw), int(bh)) == tuple(r.bbox)
        assert np.float32(float(score)) == np.float32(r.score)
