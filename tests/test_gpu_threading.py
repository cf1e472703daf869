"""The threading row of the drop-in boundary: the reference constructs the
tracker on the main thread (/root/reference/src/main.rs:49 -> src/pipeline_ir.rs:89) and uses it only
on the GStreamer streaming thread, serialised by a Mutex (src/pipeline.rs:55-67,110-119) => the handle
must be `Send`: no thread affinity, every entry point selects its own device and restores the
caller's. Tested: a tracker created on one thread and driven from another; two trackers and two
groups driven CONCURRENTLY from two threads - every result bit-identical to the single-threaded run;
the calling thread's current HIP device unchanged after every call; the same from C with pthreads."""
import os
import subprocess
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, N = 640, 480, 24


def _clip(gpu, seed):
    sc = gpu.synth.MovingSquare(W, H, 64, seed=seed)
    return sc, [np.ascontiguousarray(sc.frame_nv12(t)) for t in range(N)]


def _key(r):
    return (bool(r.success), np.float32(r.score).tobytes(), tuple(r.bbox))


def _drive(gpu, trk, sc, frames, out, dev_log=None):
    import torch
    trk.init(gpu.NV12Frame(frames[0], W, H), gpu.BBox.new(*sc.gt_box(0)))
    for t in range(N):
        out.append(_key(trk.update(gpu.NV12Frame(frames[t], W, H))))
        if dev_log is not None:
            dev_log.append(torch.cuda.current_device())


def test_tracker_created_on_main_thread_used_on_another(gpu, weights_tiny):
    sc, frames = _clip(gpu, 21)
    ref = []
    _drive(gpu, gpu.VitTrack.new(weights_tiny), sc, frames, ref)
    trk = gpu.VitTrack.new(weights_tiny)                 # constructed here ...
    got, devs = [], []
    th = threading.Thread(target=_drive, args=(gpu, trk, sc, frames, got, devs))   # ... called only there
    th.start()
    th.join()
    assert got == ref
    assert set(devs) == {0}                              # the streaming thread's current device is untouched
    # and back on the creating thread (the reference never does this, a Send handle allows it)
    again = []
    _drive(gpu, trk, sc, frames, again)
    assert again == ref
    trk.close()                                          # destroyed on the creating thread


def test_two_trackers_driven_concurrently_equal_their_single_threaded_runs(gpu, weights_tiny):
    clips = [_clip(gpu, 31), _clip(gpu, 32)]
    refs = []
    for sc, frames in clips:
        r = []
        _drive(gpu, gpu.VitTrack.new(weights_tiny), sc, frames, r)
        refs.append(r)
    assert refs[0] != refs[1]                            # different clips: a cross-talk would show
    trks = [gpu.VitTrack.new(weights_tiny) for _ in clips]
    for rep in range(3):
        outs = [[], []]
        go = threading.Barrier(2)

        def run(k):
            go.wait()
            _drive(gpu, trks[k], clips[k][0], clips[k][1], outs[k])

        th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert outs[0] == refs[0] and outs[1] == refs[1], rep


def test_two_groups_driven_concurrently_host_and_device_paths(gpu, weights_tiny):
    """two engines (vt_group), each on its own thread: one through the pipelined host ingest
    (enqueue_host / wait_next: its own copy stream), one on device-resident frames; both must equal
    their single-threaded results"""
    import torch
    B = 3
    clips = [_clip(gpu, 41), _clip(gpu, 42)]
    dclip = torch.from_numpy(np.stack(clips[1][1])).cuda()
    fb = dclip.shape[1]

    def host_run(g, out):
        sc, frames = clips[0]
        for b in range(B):
            g.init_host(b, gpu.NV12Frame(frames[0], W, H), gpu.BBox.new(*sc.gt_box(0)))
        g.enqueue_host([gpu.NV12Frame(frames[1], W, H)] * B)
        for t in range(2, N):
            g.enqueue_host([gpu.NV12Frame(frames[t], W, H)] * B)
            out.append([_key(r) for r in g.wait_next()])
        out.append([_key(r) for r in g.wait_next()])

    def dev_run(g, out):
        sc, _ = clips[1]
        fr = lambda t: gpu.frame_nv12(dclip.data_ptr() + t * fb, dclip.data_ptr() + t * fb + W * H, W, H)
        for b in range(B):
            g.init_device(b, fr(0), gpu.BBox.new(*sc.gt_box(0)))
        for t in range(1, N):
            out.append([_key(r) for r in g.update_device([fr(t)] * B)])

    refs = [[], []]
    host_run(gpu.Group(weights_tiny, n_streams=B), refs[0])
    dev_run(gpu.Group(weights_tiny, n_streams=B), refs[1])
    ga, gb = gpu.Group(weights_tiny, n_streams=B), gpu.Group(weights_tiny, n_streams=B)
    for rep in range(3):
        outs = [[], []]
        th = [threading.Thread(target=host_run, args=(ga, outs[0])), threading.Thread(target=dev_run, args=(gb, outs[1]))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert outs[0] == refs[0] and outs[1] == refs[1], rep
        assert torch.cuda.current_device() == 0


def test_c_client_with_pthreads_equals_single_threaded_run(gpu, weights_tiny, tmp_path):
    """harness/c_client `threads`: two trackers created on the main thread, each driven by its own
    pthread at the same time - line for line the output of the single-threaded `run`"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "harness", "c_client")
    assert os.path.exists(exe), "harness/c_client missing: python __graft_entry__.py"
    sc, frames = _clip(gpu, 5)
    clip = tmp_path / "clip.nv12"
    clip.write_bytes(b"".join(f.tobytes() for f in frames))
    args = [gpu.LIB_PATH, weights_tiny, str(clip), str(W), str(H), str(N)] + [str(int(v)) for v in sc.gt_box(0)]
    one = subprocess.run([exe, "run"] + args, capture_output=True, text=True)
    two = subprocess.run([exe, "threads"] + args, capture_output=True, text=True)
    assert one.returncode == 0 and two.returncode == 0, one.stderr + two.stderr
    single = one.stdout.strip().splitlines()
    lines = two.stdout.strip().splitlines()
    assert len(single) == N and len(lines) == 2 * N
    for k in range(2):
        assert [l.split(" ", 1)[1] for l in lines[k * N:(k + 1) * N]] == single
        assert all(l.startswith(f"{k} ") for l in lines[k * N:(k + 1) * N])
