"""north_star parity gate at full length: the HIP path against COMMITTED oracle
trajectories (tests/golden/traj_*.npz, forced_*.npz; generator tests/golden/make_traj.py), so the GPU
run does not pay seconds of CPU oracle per frame.

  * closed loop, 300 frames on the headline config cfg3 (and cfg2), 60 on cfg5 (4K, ViT-L/14):
    per-frame |dx|,|dy|,|dw|,|dh| <= 1 px, equal success flags, |dscore| < 0.03, and the IoU report:
    SURVEY.md section 8(d) wrote "IoU >= 0.99 per frame", which +-1 px on a 64-px box cannot
    guarantee (one coordinate off by one is IoU 0.969, two are 0.94): the test asserts what both
    statements allow together - every frame within +-1 px, MEAN IoU >= 0.99, min IoU >= 0.90 - and
    prints the minimum and the number of frames below 0.99.
  * teacher-forced (open loop) on the FIRST-GENERATION noisy head (fitted on 128 CPU samples only,
    tests/golden/head_gen1_cfg3.npz): before every frame the HIP tracker's state is overwritten with
    the state the oracle had (vt_group_set_state_box), so each frame measures the divergence of ONE
    forward pass on an ill-conditioned head instead of a trajectory that re-synchronises. Asserted:
    same argmax cell wherever the oracle's top-1/top-2 response margin is >= MARGIN_EPS, box within
    +-1 px on every frame; the margins and the disagreements below MARGIN_EPS are printed.

The oracle is parity-unpinned against the reference (no vectors exist there)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import iou

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MARGIN_EPS = 0.01      # response units (sigmoid * hann, in [0, 1])


def _sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def _fixture(name):
    path = os.path.join(GOLD, name)
    assert os.path.exists(path), f"{name} missing: python tests/golden/make_traj.py (see its docstring)"
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def _clip(gpu, fx):
    return gpu.synth.MovingSquare(int(fx["frame_w"]), int(fx["frame_h"]), int(fx["square"]),
                                  seed=int(fx["seed"]))


@pytest.mark.parametrize("name", ["traj_cfg3_300.npz", "traj_cfg2_300.npz", "traj_cfg5_60.npz"])
def test_closed_loop_trajectory_vs_committed_oracle(gpu, name, capsys):
    fx = _fixture(name)
    cfg = str(fx["config"])
    weights = gpu.weights.ensure_weights(cfg)
    assert _sha256(weights) == str(fx["weights_sha256"]), "fixture was made with other weights"
    sc = _clip(gpu, fx)
    w, h, n = sc.w, sc.h, int(fx["frames"])
    trk = gpu.VitTrack(weights)
    boxes, scores, succ = [], [], []
    for t in range(n):
        f = gpu.NV12Frame(sc.frame_nv12(t), w, h)
        if t == 0:      # init then update on the SAME frame (src/tracker_context.rs:88-90)
            trk.init(f, gpu.BBox.new(*sc.gt_box(0)))
        r = trk.update(f)
        boxes.append(r.bbox)
        scores.append(r.score)
        succ.append(int(r.success))
    boxes = np.array(boxes)
    d = np.abs(boxes - fx["bbox"])
    ious = np.array([iou(tuple(a), tuple(b)) for a, b in zip(boxes, fx["bbox"])])
    gt_iou = np.array([iou(tuple(a), tuple(b)) for a, b in zip(fx["bbox"], fx["gt"])])
    with capsys.disabled():
        print(f"\n[{name}] {n} frames: max |delta| {d.max()} px, IoU(hip, oracle) min {ious.min():.4f} "
              f"mean {ious.mean():.5f}, frames below 0.99: {(ious < 0.99).sum()}, identical boxes: "
              f"{(d.max(axis=1) == 0).sum()}; oracle vs ground truth min IoU {gt_iou.min():.3f}; "
              f"min top-1/top-2 margin {fx['margin'].min():.4f}")
    assert d.max() <= 1, f"max |delta| = {d.max()} px at frame {int(d.max(axis=1).argmax())}"
    assert ious.mean() >= 0.99 and ious.min() >= 0.90
    assert np.array_equal(np.array(succ), fx["success"].astype(int)), "success flags differ"
    assert np.abs(np.array(scores) - fx["score"]).max() < 0.03
    assert gt_iou.min() > 0.5, "the oracle lost the target: the parity above would be vacuous"


def test_teacher_forced_on_the_noisy_first_generation_head(gpu, capsys):
    fx = _fixture("forced_cfg3_300.npz")
    with np.load(os.path.join(GOLD, "head_gen1_cfg3.npz")) as z:
        head = {k: z[k] for k in z.files}
    weights = gpu.weights.ensure_weights(
        "cfg3", path=os.path.join(gpu.weights.default_cache_dir(), "vitb16_t192_s384_gen1head.vtw"),
        head=head)
    assert _sha256(weights) == str(fx["weights_sha256"]), "fixture was made with other weights"
    sc = _clip(gpu, fx)
    w, h, n = sc.w, sc.h, int(fx["frames"])
    trk = gpu.VitTrack(weights)
    g = trk.as_group()
    idx, boxes, scores = [], [], []
    for t in range(n):
        f = gpu.NV12Frame(sc.frame_nv12(t), w, h)
        if t == 0:
            trk.init(f, gpu.BBox.new(*sc.gt_box(0)))
        g.set_state_box(0, fx["state"][t])          # the oracle's state before its update of frame t
        r = trk.update(f)
        idx.append(g.read_state()["last_idx"])
        boxes.append(r.bbox)
        scores.append(r.score)
    idx, boxes = np.array(idx), np.array(boxes)
    d = np.abs(boxes - fx["bbox"])
    clear = fx["margin"] >= MARGIN_EPS
    differ = idx != fx["idx"]
    swapped = differ & (idx == fx["idx2"])           # HIP took the oracle's runner-up
    with capsys.disabled():
        print(f"\n[teacher-forced, gen-1 head] {n} frames: oracle margin min {fx['margin'].min():.5f} "
              f"median {np.median(fx['margin']):.4f}, frames with margin < {MARGIN_EPS}: {(~clear).sum()}; "
              f"argmax differs on {differ.sum()} frames ({swapped.sum()} of them = the oracle's "
              f"runner-up), all with margin <= {fx['margin'][differ].max() if differ.any() else 0:.5f}; "
              f"max |delta box| {d.max()} px, max |delta score| "
              f"{np.abs(np.array(scores) - fx['score']).max():.4f}; oracle success on "
              f"{int(fx['success'].sum())} frames")
    assert not (differ & clear).any(), \
        f"argmax differs at margin {fx['margin'][differ & clear].max():.4f} (frame {int(np.argmax(differ & clear))})"
    assert d.max() <= 1, f"open-loop box differs by {d.max()} px at frame {int(d.max(axis=1).argmax())}"
    assert np.abs(np.array(scores) - fx["score"]).max() < 0.03
