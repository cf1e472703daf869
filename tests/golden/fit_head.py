"""Fits the centre head of a synthetic model and writes it to
gstreamer-vit-tracker_amd/assets/head_<config>.npz  (the committed, trained part of the weights).

Why this exists: the reference's network (object_tracking_vittrack_2023sep.rknn,
/root/reference/src/main.rs:25) is not available, so the encoder weights are a seeded hash
(gstreamer-vit-tracker_amd/weights.py). A random head on top of them produces a flat score map,
and a closed-loop HIP-vs-oracle parity test on such a map would hinge on argmax ties. This script
trains only the small convolutional head, on synthetic moving-square crops, against features that
the CPU oracle's encoder computes from exactly those seeded weights.

Two ways to get the encoder features the head is trained on:
  CPU (oracle):  python tests/golden/fit_head.py tiny cfg2 cfg3 [cfg5]            (minutes, ~100 samples)
  GPU (the HIP path's own final-LayerNorm tap, torch on the GPU for the fit):
                 python tests/golden/fit_head.py --gpu --samples 4000 --steps 4000 --out DIR cfg3
The committed assets come from the GPU mode (40x the data): a head fitted on ~100 samples predicts
the box with 1-2 px of frame-to-frame noise, which makes two implementations that differ by one
rounding flip drift apart for a few frames. Deterministic up to summation order (seeds fixed).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import gstreamer_vit_tracker_amd as vt  # noqa: E402
from oracle import vit_ref as R  # noqa: E402


def make_samples(model: R.Model, n: int, seed: int, w=960, h=540):
    """-> feats [n, Ns, D] (bf16-valued f32), targets [n, 4] = (cxn, cyn, wn, hn) in crop units"""
    rng = np.random.default_rng(seed)
    feats, targets = [], []
    hdr = model.hdr
    for i in range(n):
        sw = int(rng.integers(36, 120))
        sh = sw if rng.random() < 0.6 else int(rng.integers(36, 120))
        sc = vt.synth.MovingSquare(w, h, square=max(sw, sh), seed=int(rng.integers(1 << 30)))
        # rectangle: crop the square texture
        x = int(rng.integers(0, w - sw))
        y = int(rng.integers(0, h - sh))
        yy = sc.bg_y.copy()
        yy[y:y + sh, x:x + sw] = sc.sq_y[:sh, :sw]
        uv = np.full(((h + 1) // 2, w // 2, 2), 128, np.uint8)
        uv[y // 2:(y + sh + 1) // 2, x // 2:(x + sw + 1) // 2, 0] = sc.sq_u
        uv[y // 2:(y + sh + 1) // 2, x // 2:(x + sw + 1) // 2, 1] = sc.sq_v
        buf = np.concatenate([yy.reshape(-1), uv.reshape(-1)])
        fr = R.Frame.nv12(buf, w, h)
        gt = np.array([x, y, sw, sh], np.float32)
        # template from the true box; search window from a jittered "previous" box
        tpl = R.preproc(fr, gt, 2.0, model.T, model.patch, model.kpad, hdr["norm_a"],
                        hdr["norm_b"])
        jit = rng.uniform(-0.22, 0.22, 2) * 4.0 * np.sqrt(sw * sh) * (rng.random() < 0.85)
        sj = np.exp(rng.uniform(-0.12, 0.12, 2))
        pw, ph = sw * sj[0], sh * sj[1]
        prev = np.array([x + sw / 2 + jit[0] - pw / 2, y + sh / 2 + jit[1] - ph / 2, pw, ph],
                        np.float32)
        geo = R.crop_geometry(prev, 4.0, model.S)
        srch = R.preproc(fr, prev, 4.0, model.S, model.patch, model.kpad, hdr["norm_a"],
                         hdr["norm_b"])
        out = model.forward(np.concatenate([tpl, srch], axis=0))
        feats.append(out["feat"])
        side = geo[3]
        targets.append([(x + sw / 2 - (geo[0] + 0.5)) / side, (y + sh / 2 - (geo[1] + 0.5)) / side,
                        sw / side, sh / side])
    return np.stack(feats).astype(np.float32), np.array(targets, np.float32)


def unfold3x3(t):
    """t [n, g, g, c] -> [n, g, g, 9c], column (ky*3+kx)*c + ci, zero padding — the same im2col the
    HIP head uses, so the 3x3 convolutions run as plain matrix products (fast on CPU BLAS)"""
    n, g, _, c = t.shape
    p = F.pad(t, (0, 0, 1, 1, 1, 1))
    return torch.cat([p[:, ky:ky + g, kx:kx + g, :] for ky in range(3) for kx in range(3)], dim=-1)


class Head(torch.nn.Module):
    """channels-last centre head: 1x1 conv, three 3x3 convs (as im2col + linear), 1x1 conv to 5"""

    def __init__(self, d, c):
        super().__init__()
        self.c0 = torch.nn.Linear(d, c)
        self.c1 = torch.nn.Linear(9 * c, c)
        self.c2 = torch.nn.Linear(9 * c, c)
        self.c3 = torch.nn.Linear(9 * c, c)
        self.c4 = torch.nn.Linear(c, 5)

    def forward(self, x):                     # x [n, g, g, d]
        x = F.relu(self.c0(x))
        x = F.relu(self.c1(unfold3x3(x)))
        x = F.relu(self.c2(unfold3x3(x)))
        x = F.relu(self.c3(unfold3x3(x)))
        return self.c4(x).permute(0, 3, 1, 2)  # [n, 5, g, g]


def fit(cfg_name: str, n_train: int, steps: int, out_path: str | None = None):
    cfg = vt.weights.get_config(cfg_name)
    torch.manual_seed(0)
    t0 = time.time()
    blob = vt.weights.pack_blob(cfg, vt.weights.generate_tensors(cfg, use_asset=False))
    model = R.Model(blob)
    cache = os.path.join(os.environ.get("VT_FIT_CACHE", "/tmp/vt_fit_cache"),
                         f"{cfg.name}_{n_train}.npz")
    if os.path.exists(cache):
        with np.load(cache) as z:
            feats, tg = z["feats"], z["tg"]
    else:
        feats, tg = make_samples(model, n_train, seed=1234)
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        np.savez(cache, feats=feats, tg=tg)
    g, d, c = model.gs, model.D, model.C
    print(f"[{cfg.name}] features {feats.shape} in {time.time() - t0:.1f}s", flush=True)
    x = torch.from_numpy(feats).reshape(n_train, g, g, d).contiguous()
    tgt = torch.from_numpy(tg)
    # labels
    cx, cy = tgt[:, 0] * g, tgt[:, 1] * g
    ix = cx.floor().clamp(0, g - 1).long()
    iy = cy.floor().clamp(0, g - 1).long()
    gx = torch.arange(g).float() + 0.5
    heat = torch.exp(-((gx[None, None, :] - cx[:, None, None]) ** 2 +
                       (gx[None, :, None] - cy[:, None, None]) ** 2) / (2 * 0.65 ** 2))
    inside = ((tgt[:, 0] > 0) & (tgt[:, 0] < 1) & (tgt[:, 1] > 0) & (tgt[:, 1] < 1)).float()
    head = Head(d, c)
    opt = torch.optim.Adam(head.parameters(), lr=2e-3)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-3, total_steps=steps)
    ar = torch.arange(n_train)
    for step in range(steps):
        out = head(x)
        score_loss = F.binary_cross_entropy_with_logits(out[:, 0], heat * inside[:, None, None])
        # every cell of the 3x3 window around the positive cell regresses the SAME centre (offset
        # range [-1, 2] cells: 3*sigmoid - 1) and the same size, so the decode's window mean is
        # insensitive to which of two near-tied cells wins the argmax
        loss_reg = 0.0
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                jy, jx = (iy + dy).clamp(0, g - 1), (ix + dx).clamp(0, g - 1)
                q = out[ar, :, jy, jx]
                tx = (cx - jx.float()).clamp(-0.97, 1.97)
                ty = (cy - jy.float()).clamp(-0.97, 1.97)
                wgt = inside * (1.0 if (dx == 0 and dy == 0) else 0.5)
                loss_reg = loss_reg + (
                    F.l1_loss(3 * torch.sigmoid(q[:, 1]) - 1, tx, reduction="none") * wgt).mean() + (
                    F.l1_loss(3 * torch.sigmoid(q[:, 2]) - 1, ty, reduction="none") * wgt).mean() + (
                    F.l1_loss(torch.sigmoid(q[:, 3]), tgt[:, 2], reduction="none") * wgt).mean() + (
                    F.l1_loss(torch.sigmoid(q[:, 4]), tgt[:, 3], reduction="none") * wgt).mean()
        loss = 20.0 * score_loss + loss_reg
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if step % 50 == 0 or step == steps - 1:
            print(f"[{cfg.name}] step {step} loss {loss.item():.4f} score {score_loss.item():.4f} "
                  f"reg {float(loss_reg.detach()):.4f} ({time.time() - t0:.0f}s)", flush=True)

    w4 = np.zeros((8, c), np.float32)
    b4 = np.zeros((1, 8), np.float32)
    w4[:5] = head.c4.weight.detach().numpy()
    b4[0, :5] = head.c4.bias.detach().numpy()
    asset = {"head.w4": w4, "head.b4": b4}
    for k, lin in enumerate((head.c0, head.c1, head.c2, head.c3)):
        asset[f"head.w{k}"] = lin.weight.detach().numpy().copy()
        asset[f"head.b{k}"] = lin.bias.detach().numpy().reshape(1, c).copy()
    # the bf16 tensors are stored as the bf16-rounded values so the asset is what the blob holds
    for k in vt.weights.HEAD_BF16:
        asset[k] = R.bf16r(asset[k].astype(np.float32))
    path = out_path or vt.weights.head_asset_path(cfg)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **{k: v.astype(np.float32) for k, v in asset.items()})
    print(f"[{cfg.name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)
    if out_path is None:
        validate(cfg_name)
    return asset


def make_samples_gpu(cfg, n: int, seed: int, w=960, h=540, bs=32, sq_range=(36, 120)):
    """features from the HIP path: for each sample the template comes from the true box, the search
    window from a jittered previous box (vt_group_set_state_box), feat = final-LN tap"""
    wpath = vt.weights.ensure_weights(cfg.name, use_asset=False)
    grp = vt.Group(wpath, n_streams=bs)
    rng = np.random.default_rng(seed)
    bgs = [rng.integers(44, 77, size=(h, w), dtype=np.uint8) for _ in range(8)]
    feats = torch.empty((n, cfg.n_s, cfg.dim), dtype=torch.bfloat16, device="cuda")
    targets = np.zeros((n, 4), np.float32)
    fbytes = w * h * 3 // 2
    for b0 in range(0, n, bs):
        nb = min(bs, n - b0)
        host = np.empty((bs, fbytes), np.uint8)
        gts, prevs = [], []
        for i in range(bs):
            sw = int(rng.integers(*sq_range))
            sh = sw if rng.random() < 0.6 else int(rng.integers(*sq_range))
            x, y = int(rng.integers(0, w - sw)), int(rng.integers(0, h - sh))
            yy = bgs[int(rng.integers(8))].copy()
            yy[y:y + sh, x:x + sw] = (200 + rng.integers(-12, 13, size=(sh, sw))).astype(np.uint8)
            uv = np.full((h // 2, w // 2, 2), 128, np.uint8)
            uv[y // 2:(y + sh + 1) // 2, x // 2:(x + sw + 1) // 2, 0] = 90
            uv[y // 2:(y + sh + 1) // 2, x // 2:(x + sw + 1) // 2, 1] = 200
            host[i, : w * h] = yy.reshape(-1)
            host[i, w * h:] = uv.reshape(-1)
            jit = rng.uniform(-0.22, 0.22, 2) * 4.0 * np.sqrt(sw * sh) * (rng.random() < 0.85)
            sj = np.exp(rng.uniform(-0.12, 0.12, 2))
            pw, ph = sw * sj[0], sh * sj[1]
            # the tracker's state is an integer box
            prev = np.floor(np.array([x + sw / 2 + jit[0] - pw / 2, y + sh / 2 + jit[1] - ph / 2,
                                      pw, ph]) + 0.5).astype(np.float32)
            prev[2:] = np.maximum(prev[2:], 10)
            gts.append((x, y, sw, sh))
            prevs.append(prev)
        dev = torch.from_numpy(host).cuda()
        frames = [vt.frame_nv12(dev[i].data_ptr(), dev[i].data_ptr() + w * h, w, h) for i in range(bs)]
        for i in range(bs):
            grp.init_device(i, frames[i], vt.BBox.new(*gts[i]))
            grp.set_state_box(i, prevs[i])
        grp.update_device(frames)
        for i in range(nb):
            f = grp.read_tensor("feat", i).reshape(cfg.n_s, cfg.dim)
            feats[b0 + i] = torch.from_numpy(f).cuda().to(torch.bfloat16)
            geo = R.crop_geometry(prevs[i], 4.0, cfg.search)
            x, y, sw, sh = gts[i]
            side = geo[3]
            targets[b0 + i] = [(x + sw / 2 - (geo[0] + 0.5)) / side,
                               (y + sh / 2 - (geo[1] + 0.5)) / side, sw / side, sh / side]
    return feats, targets


def head_loss(out, heat, inside, cx, cy, ix, iy, tgt, g, size_weight=1.0):
    ar = torch.arange(out.shape[0], device=out.device)
    score_loss = F.binary_cross_entropy_with_logits(out[:, 0], heat * inside[:, None, None])
    loss_reg = 0.0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            jy, jx = (iy + dy).clamp(0, g - 1), (ix + dx).clamp(0, g - 1)
            q = out[ar, :, jy, jx]
            tx = (cx - jx.float()).clamp(-0.97, 1.97)
            ty = (cy - jy.float()).clamp(-0.97, 1.97)
            wgt = inside * (1.0 if (dx == 0 and dy == 0) else 0.5)
            loss_reg = loss_reg + (
                F.l1_loss(3 * torch.sigmoid(q[:, 1]) - 1, tx, reduction="none") * wgt).mean() + (
                F.l1_loss(3 * torch.sigmoid(q[:, 2]) - 1, ty, reduction="none") * wgt).mean() + size_weight * ((
                F.l1_loss(torch.sigmoid(q[:, 3]), tgt[:, 2], reduction="none") * wgt).mean() + (
                F.l1_loss(torch.sigmoid(q[:, 4]), tgt[:, 3], reduction="none") * wgt).mean())
    return 20.0 * score_loss + loss_reg, score_loss, loss_reg


def export_head(head, cfg, out_dir=None):
    c = cfg.head_ch
    head = head.cpu()
    w4 = np.zeros((8, c), np.float32)
    b4 = np.zeros((1, 8), np.float32)
    w4[:5] = head.c4.weight.detach().numpy()
    b4[0, :5] = head.c4.bias.detach().numpy()
    asset = {"head.w4": w4, "head.b4": b4}
    for k, lin in enumerate((head.c0, head.c1, head.c2, head.c3)):
        asset[f"head.w{k}"] = lin.weight.detach().numpy().copy()
        asset[f"head.b{k}"] = lin.bias.detach().numpy().reshape(1, c).copy()
    for k in vt.weights.HEAD_BF16:
        asset[k] = R.bf16r(asset[k].astype(np.float32))
    path = vt.weights.head_asset_path(cfg)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, os.path.basename(path))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **{k: v.astype(np.float32) for k, v in asset.items()})
    print(f"[{cfg.name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)
    return asset


def fit_gpu(cfg_name: str, n_train: int, steps: int, out_dir: str, batch: int = 256, noise: float = 0.0,
            frame=(960, 540), sq_range=(36, 120), size_weight: float = 1.0):
    """noise > 0: Gaussian noise of that standard deviation on every feature element of every step. The
    two bf16 implementations that the parity tests compare differ by about 4e-3 of the feature maximum
    (a few 1e-3 per element, DESIGN.md section 5 / tools/arbiter.py); a head fitted on clean features
    only may turn that into a pixel of box difference, one that has seen such perturbations does not."""
    cfg = vt.weights.get_config(cfg_name)
    torch.manual_seed(0)
    t0 = time.time()
    feats, tg = make_samples_gpu(cfg, n_train, seed=1234, w=frame[0], h=frame[1], sq_range=sq_range)
    g, d, c = cfg.grid_s, cfg.dim, cfg.head_ch
    print(f"[{cfg.name}] {n_train} samples of HIP features in {time.time() - t0:.1f}s", flush=True)
    tgt = torch.from_numpy(tg).cuda()
    cx, cy = tgt[:, 0] * g, tgt[:, 1] * g
    ix = cx.floor().clamp(0, g - 1).long()
    iy = cy.floor().clamp(0, g - 1).long()
    gx = torch.arange(g, device="cuda").float() + 0.5
    inside = ((tgt[:, 0] > 0) & (tgt[:, 0] < 1) & (tgt[:, 1] > 0) & (tgt[:, 1] < 1)).float()
    head = Head(d, c).cuda()
    opt = torch.optim.Adam(head.parameters(), lr=2e-3)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-3, total_steps=steps)
    gen = torch.Generator(device="cuda").manual_seed(1)
    for step in range(steps):
        idx = torch.randint(0, n_train, (min(batch, n_train),), device="cuda", generator=gen)
        x = feats[idx].float().reshape(-1, g, g, d)
        if noise > 0:
            x = x + noise * torch.randn(x.shape, device="cuda", generator=gen)
        heat = torch.exp(-((gx[None, None, :] - cx[idx, None, None]) ** 2 +
                           (gx[None, :, None] - cy[idx, None, None]) ** 2) / (2 * 0.65 ** 2))
        loss, sl, rl = head_loss(head(x), heat, inside[idx], cx[idx], cy[idx], ix[idx], iy[idx],
                                 tgt[idx], g, size_weight)
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if step % 250 == 0 or step == steps - 1:
            print(f"[{cfg.name}] step {step} loss {loss.item():.4f} score {sl.item():.4f} "
                  f"reg {float(rl.detach()):.4f} ({time.time() - t0:.0f}s)", flush=True)
    asset = export_head(head, cfg, out_dir)
    validate_gpu(cfg, asset)


def validate_gpu(cfg, asset, frames: int = 200):
    """closed loop of the HIP path on a held-out clip: IoU vs ground truth"""
    wpath = vt.weights.ensure_weights(cfg.name, path=f"/tmp/vt_fit_{cfg.name}.vtw", head=asset)
    big = cfg.search >= 256
    w, h, sq = (3840, 2160, 160) if cfg.patch == 14 else ((1920, 1080, 64) if big else (640, 480, 64))
    sc = vt.synth.MovingSquare(w, h, sq, seed=0)
    trk = vt.VitTrack(wpath)
    ious, scores, errs = [], [], []
    g = trk.as_group()
    for t in range(frames):
        fr = vt.NV12Frame(sc.frame_nv12(t), w, h)
        if t == 0:
            trk.init(fr, vt.BBox.new(*sc.gt_box(0)))
        r = trk.update(fr)
        gx, gy, gw, gh = sc.gt_box(t)
        bx, by, bw, bh = r.bbox
        iw = max(0, min(gx + gw, bx + bw) - max(gx, bx))
        ih = max(0, min(gy + gh, by + bh) - max(gy, by))
        ious.append(iw * ih / (gw * gh + bw * bh - iw * ih))
        scores.append(r.score)
        fb = g.read_state()["last_fbox"]
        errs.append([fb[0] + fb[2] / 2 - (gx + gw / 2), fb[1] + fb[3] / 2 - (gy + gh / 2), fb[2] - gw, fb[3] - gh])
    e = np.array(errs)
    print(f"[{cfg.name}] HIP closed loop {frames} frames: IoU vs GT min {min(ious):.3f} mean "
          f"{np.mean(ious):.3f}; score min {min(scores):.3f}; float box error vs GT (cx, cy, w, h) px: mean "
          f"{np.round(e.mean(axis=0), 2).tolist()} std {np.round(e.std(axis=0), 2).tolist()} max |.| "
          f"{np.round(np.abs(e).max(axis=0), 2).tolist()}", flush=True)


def validate(cfg_name: str, frames: int = 40):
    """closed loop on a held-out clip with the oracle: IoU vs ground truth and score margin"""
    cfg = vt.weights.get_config(cfg_name)
    wpath = vt.weights.ensure_weights(cfg_name, force=True)
    trk = R.VitTrackRef(wpath)
    big = cfg.search >= 256
    w, h = (1920, 1080) if big else (640, 480)
    sc = vt.synth.MovingSquare(w, h, 64, seed=0)
    ious, scores, margins = [], [], []
    for t in range(frames):
        fr = R.Frame.nv12(sc.frame_nv12(t), w, h)
        if t == 0:
            trk.init(fr, sc.gt_box(0))
        r = trk.update(fr, taps=True)
        gx, gy, gw, gh = sc.gt_box(t)
        bx, by, bw, bh = r.bbox
        iw = max(0, min(gx + gw, bx + bw) - max(gx, bx))
        ih = max(0, min(gy + gh, by + bh) - max(gy, by))
        ious.append(iw * ih / (gw * gh + bw * bh - iw * ih))
        scores.append(r.score)
        s = 1 / (1 + np.exp(-trk.last["head_out"][:, 0])) * trk.m.t["hann"].reshape(-1)
        top = np.sort(s)[-2:]
        margins.append(float(top[1] - top[0]))
    print(f"[{cfg.name}] validate: IoU vs GT min {min(ious):.3f} mean {np.mean(ious):.3f}; score "
          f"min {min(scores):.3f}; top1-top2 response margin min {min(margins):.3f} "
          f"median {np.median(margins):.3f}", flush=True)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["tiny"])
    ap.add_argument("--gpu", action="store_true", help="features from the HIP path, fit with torch on the GPU")
    ap.add_argument("--samples", type=int, default=4000)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--out", default=None, help="directory for the .npz (default: the package's assets/)")
    ap.add_argument("--noise", type=float, default=0.0, help="--gpu: std of Gaussian feature noise per step")
    ap.add_argument("--frame", default="960x540", help="--gpu: size of the synthetic training frames")
    ap.add_argument("--squares", default="36-120", help="--gpu: range of target sizes in pixels")
    ap.add_argument("--size-weight", type=float, default=1.0, help="--gpu: weight of the two size terms of the loss")
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    torch.set_num_threads(int(os.environ.get("FIT_THREADS", "8")))
    for nme in a.configs:
        if nme.startswith("validate:"):
            validate(nme.split(":", 1)[1])
        elif a.gpu:
            fw_, fh_ = (int(v) for v in a.frame.split("x"))
            lo_, hi_ = (int(v) for v in a.squares.split("-"))
            fit_gpu(nme, a.samples, a.steps, a.out, batch=a.batch, noise=a.noise, frame=(fw_, fh_), sq_range=(lo_, hi_),
                    size_weight=a.size_weight)
        else:
            cfgs = {"tiny": (256, 500), "cfg2": (160, 400), "cfg3": (128, 1000), "cfg5": (64, 700)}
            n, st = cfgs.get(nme, (128, 400))
            fit(nme, n, st)
