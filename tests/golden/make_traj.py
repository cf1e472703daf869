"""Generates the committed oracle trajectories the GPU parity tests compare against
(tests/test_gpu_trajectories.py), so that the GPU run does not pay seconds of CPU oracle per frame.

Everything here runs the CPU oracle only (oracle/vit_ref.py + oracle/vt_oracle.c); nothing is read
from /root/reference (it holds no vectors for this path, SURVEY.md §8c: PARITY UNPINNED).

  python tests/golden/make_traj.py traj cfg3 --frames 300 --seed 1
      closed loop of the oracle on the synthetic clip -> tests/golden/traj_<cfg>_<frames>.npz
      (per frame: box, score, success, argmax cell, top-1/top-2 response margin, state box)
  python tests/golden/make_traj.py traj cfg3 --frames 300 --seed 9 --square 80 --tag _b
      a second clip (other background, phase and target size) -> traj_cfg3_300_b.npz
  python tests/golden/make_traj.py gen1head cfg3
      the FIRST-GENERATION head: fitted on ~128 CPU-oracle samples only (noisy, 1-2 px
      of frame-to-frame jitter) -> tests/golden/head_gen1_<cfg>.npz
  python tests/golden/make_traj.py forced cfg3 --frames 300 --seed 5
      closed loop of the oracle WITH THE GEN-1 HEAD; the state box the oracle had before every
      update is recorded so that the HIP path can be teacher-forced (vt_group_set_state_box) and
      compared frame by frame, open loop -> tests/golden/forced_<cfg>_<frames>.npz

  python tests/golden/make_traj.py fbox traj_cfg3_300.npz
      adds `fbox` - the oracle's FLOAT box of every update (x, y, w, h before the integer rounding) - to
      an existing fixture: every frame is re-evaluated from the fixture's own `state` (one forward pass
      each, the oracle is deterministic), the stored integer box / score / cell must come out identical,
      and the float box is appended. The teacher-forced GPU tests assert |HIP float box - fbox| on it.

Each file records the SHA-256 of the weight blob it was made with; the tests rebuild the blob and
refuse to compare against a fixture made from other weights.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import gstreamer_vit_tracker_amd as vt  # noqa: E402  (weights writer + synthetic clip only)
from oracle import vit_ref as R  # noqa: E402

CLIPS = {  # config -> (frame w, h, square): BASELINE.json configs[1], [2], [4]
    "cfg2": (1920, 1080, 64), "cfg3": (1920, 1080, 64), "cfg5": (3840, 2160, 160),
    "tiny": (640, 480, 64),
}


def sha256_file(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def gen1_head_path(cfg: str) -> str:
    return os.path.join(HERE, f"head_gen1_{cfg}.npz")


def gen1_weights(cfg: str) -> str:
    """blob = seeded encoder + the committed gen-1 head"""
    with np.load(gen1_head_path(cfg)) as z:
        head = {k: z[k] for k in z.files}
    path = os.path.join(vt.weights.default_cache_dir(), f"{vt.weights.get_config(cfg).name}_gen1head.vtw")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    return vt.weights.ensure_weights(cfg, path=path, head=head)


def run(cfg: str, weights: str, frames: int, seed: int, out: str, verbose=True, square: int = 0, hide=None):
    w, h, sq = CLIPS[cfg]
    sq = square or sq
    sc = vt.synth.MovingSquare(w, h, sq, seed=seed, hide=hide)
    trk = R.VitTrackRef(weights)
    hann = trk.m.t["hann"].reshape(-1)
    rec = {k: [] for k in ("state", "bbox", "score", "success", "idx", "idx2", "margin", "gt", "fbox")}
    t0 = time.time()
    for t in range(frames):
        fr = R.Frame.nv12(sc.frame_nv12(t), w, h)
        if t == 0:     # init then update on the same frame (src/tracker_context.rs:88-90)
            trk.init(fr, sc.gt_box(0))
        rec["state"].append(trk.box.copy())
        r = trk.update(fr, taps=True)
        resp = (1.0 / (1.0 + np.exp(-trk.last["head_out"][:, 0].astype(np.float64)))) * hann
        order = np.argsort(-resp, kind="stable")
        rec["bbox"].append(r.bbox)
        rec["fbox"].append(r.fbox)
        rec["score"].append(r.score)
        rec["success"].append(int(r.success))
        rec["idx"].append(r.idx)
        rec["idx2"].append(int(order[1]))
        rec["margin"].append(float(resp[order[0]] - resp[order[1]]))
        rec["gt"].append(sc.gt_box(t))
        if verbose and (t % 20 == 0 or t == frames - 1):
            print(f"[{cfg}] frame {t}: {r} idx {r.idx} margin {rec['margin'][-1]:.4f} "
                  f"({time.time() - t0:.0f}s)", flush=True)
    np.savez_compressed(
        out, config=cfg, frame_w=w, frame_h=h, square=sq, seed=seed, frames=frames,
        weights_sha256=sha256_file(weights), hide=np.array(hide if hide else (0, 0), np.int32),
        state=np.array(rec["state"], np.float32), bbox=np.array(rec["bbox"], np.int32),
        score=np.array(rec["score"], np.float32), success=np.array(rec["success"], np.int8),
        idx=np.array(rec["idx"], np.int32), idx2=np.array(rec["idx2"], np.int32),
        margin=np.array(rec["margin"], np.float32), gt=np.array(rec["gt"], np.int32),
        fbox=np.array(rec["fbox"], np.float32))
    print(f"wrote {out} ({os.path.getsize(out)} bytes)", flush=True)


def add_fbox(name: str):
    """re-evaluate every frame of an existing fixture from its recorded state; verify; append `fbox`"""
    path = os.path.join(HERE, name)
    with np.load(path) as z:
        fx = {k: z[k] for k in z.files}
    cfg = str(fx["config"])
    weights = gen1_weights(cfg) if name.startswith("forced_") else vt.weights.ensure_weights(cfg)
    assert sha256_file(weights) == str(fx["weights_sha256"]), "fixture was made with other weights"
    hide = tuple(int(v) for v in fx["hide"]) if "hide" in fx and fx["hide"][1] > fx["hide"][0] else None
    w, h = int(fx["frame_w"]), int(fx["frame_h"])
    sc = vt.synth.MovingSquare(w, h, int(fx["square"]), seed=int(fx["seed"]), hide=hide)
    trk = R.VitTrackRef(weights)
    fbox, t0 = [], time.time()
    for t in range(int(fx["frames"])):
        fr = R.Frame.nv12(sc.frame_nv12(t), w, h)
        if t == 0:
            trk.init(fr, sc.gt_box(0))
        trk.box = fx["state"][t].astype(np.float32).copy()
        r = trk.update(fr)
        assert tuple(r.bbox) == tuple(int(v) for v in fx["bbox"][t]) and r.idx == int(fx["idx"][t]) and \
            np.float32(r.score) == fx["score"][t] and int(r.success) == int(fx["success"][t]), \
            f"frame {t}: the oracle no longer reproduces the fixture ({r} idx {r.idx})"
        fbox.append(r.fbox)
        if t % 20 == 0:
            print(f"[{name}] frame {t} fbox {r.fbox} ({time.time() - t0:.0f}s)", flush=True)
    fx["fbox"] = np.array(fbox, np.float32)
    np.savez_compressed(path, **fx)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "fbox":
        add_fbox(sys.argv[2])
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["traj", "gen1head", "forced"])
    ap.add_argument("cfg", choices=sorted(CLIPS))
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--square", type=int, default=0, help="target side in px (default: the configuration's clip)")
    ap.add_argument("--tag", default="", help="suffix of the output file, e.g. _b for a second clip")
    ap.add_argument("--hide", type=int, nargs=2, default=None, metavar=("T0", "T1"),
                    help="occlusion: the target is absent from frames T0 <= t < T1")
    a = ap.parse_args()
    if a.what == "traj":
        run(a.cfg, vt.weights.ensure_weights(a.cfg), a.frames, a.seed,
            os.path.join(HERE, f"traj_{a.cfg}_{a.frames}{a.tag}.npz"), square=a.square, hide=a.hide)
    elif a.what == "gen1head":
        import importlib.util
        spec = importlib.util.spec_from_file_location("fit_head", os.path.join(HERE, "fit_head.py"))
        fh = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(fh)
        import torch
        torch.set_num_threads(int(os.environ.get("FIT_THREADS", "4")))
        n, st = {"tiny": (256, 500), "cfg2": (160, 400), "cfg3": (128, 1000), "cfg5": (64, 700)}[a.cfg]
        fh.fit(a.cfg, n, st, out_path=gen1_head_path(a.cfg))
    else:
        run(a.cfg, gen1_weights(a.cfg), a.frames, a.seed,
            os.path.join(HERE, f"forced_{a.cfg}_{a.frames}.npz"))
