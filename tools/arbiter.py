"""Arbiter for HIP-vs-oracle disagreements: who is further from the un-quantised answer?

  python tools/arbiter.py cfg5 --frames 16 [--out profiles/r03_arbiter_cfg5.txt]

The parity tests compare two bf16 implementations of this build's network - the HIP kernels and
oracle/vit_ref.py - written by the same hand with mirrored rounding points. Where their boxes differ by a
pixel, neither is "right" by construction. This tool measures both against a third, independent,
UN-quantised formulation (oracle/torch_ref.py: torch.nn.functional, float64, no bf16 anywhere):

  per frame of the synthetic clip, teacher-forced (the HIP tracker's state is overwritten with the
  oracle's before the update, so all three see bit-identical patch rows):
     distance to truth of the oracle and of the HIP taps at layer<first, mid, last>, feat, head logits,
     the float box each one decodes to (same decode: oracle/vt_oracle.c vto_decode on each one's logits),
     and the top-1 / top-2 response margin of the truth.

Reading: HIP further from truth than the oracle at some stage -> a kernel problem at that stage; both
equally far -> the disagreement is bf16 noise, and what makes a 1 px box difference out of it is the
head's conditioning. Needs a GPU (HIP taps) and ~10 s of CPU per ViT-L frame.

Test infrastructure: the product does not import this. The reference has no vectors for this path
(PARITY UNPINNED); boundary: VitTrack::update (/root/reference/src/tracker_context.rs:120).
"""
import argparse
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt            # noqa: E402
from oracle import vit_ref as R                   # noqa: E402
from oracle import torch_ref                      # noqa: E402

CLIPS = {"cfg2": (1920, 1080, 64), "cfg3": (1920, 1080, 64), "cfg5": (3840, 2160, 160), "tiny": (640, 480, 64)}


def rel(a, b):
    """(max, rms) of a - b relative to the max / rms of b"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / np.abs(b).max()), float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean()))


def decode(head_out, hann, grid, geo, w, h):
    ho = np.ascontiguousarray(head_out, np.float32)
    dec, ib = np.zeros(6, np.float32), np.zeros(4, np.int32)
    R.lib().vto_decode(ho.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), hann.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                       grid, geo.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), w, h,
                       dec.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ib.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return dec[1:5].copy(), ib.copy(), int(dec[5])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cfg", choices=sorted(CLIPS))
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--stride", type=int, default=7, help="clip frames between two measured frames")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--weights", default=None)
    ap.add_argument("--head", default=None, help="head asset (.npz) to build the weights with instead of the committed one")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    w, h, sq = CLIPS[a.cfg]
    weights = a.weights or vt.weights.ensure_weights(a.cfg)
    if a.head:
        with np.load(a.head) as z:
            head = {k: z[k] for k in z.files}
        weights = vt.weights.ensure_weights(a.cfg, path=f"/tmp/vt_arbiter_{a.cfg}.vtw", head=head)
    sc = vt.synth.MovingSquare(w, h, sq, seed=a.seed)
    ref = R.VitTrackRef(weights)
    truth = torch_ref.TorchModel(weights)
    trk = vt.VitTrack(weights)
    g = trk.as_group()
    g.enable_taps(True)
    mi = trk.model_info()
    n, d, L, ns = mi.tokens_template + mi.tokens_search, mi.dim, mi.layers, mi.tokens_search
    hann = np.ascontiguousarray(ref.m.t["hann"].reshape(-1), np.float32)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    say(f"# arbiter {a.cfg}: {a.frames} frames (every {a.stride}th of the clip, seed {a.seed}), weights {weights}")
    say("# distances are (max, rms) relative to the truth tensor; o = oracle (bf16, NumPy), h = HIP, t = truth (float64)")
    acc = {k: [] for k in ("o_last", "h_last", "oh_last", "o_feat", "h_feat", "o_head", "h_head", "o_box", "h_box", "oh_box")}
    t0 = time.time()
    for k in range(a.frames):
        t = k * a.stride
        buf = sc.frame_nv12(t)
        fo, fg = R.Frame.nv12(buf, w, h), vt.NV12Frame(buf, w, h)
        box = sc.gt_box(t)
        if k == 0:
            ref.init(fo, box)
            trk.init(fg, vt.BBox.new(*box))
        # both start every measured frame from the ground-truth box: one forward pass each, same input
        ref.box = np.array(box, np.float32)
        g.set_state_box(0, ref.box)
        rr = ref.update(fo, taps=True)
        rg = trk.update(fg)
        o = ref.last
        assert np.array_equal(g.read_tensor("patches").reshape(n, mi.kpad), R.bf16_bits_to_f32(o["patches"])), "inputs differ"
        tr = truth.forward(o["patches"])
        hl = {f"layer{l}": g.read_tensor(f"layer{l}").reshape(n, d) for l in (0, L // 2, L - 1)}
        hfeat = g.read_tensor("feat").reshape(ns, d)
        hho = g.read_tensor("head_out").reshape(ns, 8)
        geo = np.ascontiguousarray(o["geo"], np.float32)
        tb, tib, tidx = decode(tr["head_out"], hann, mi.score_grid, geo, w, h)
        ob, oib, oidx = decode(o["head_out"], hann, mi.score_grid, geo, w, h)
        hb, hib, hidx = decode(hho, hann, mi.score_grid, geo, w, h)
        resp = (1.0 / (1.0 + np.exp(-tr["head_out"][:, 0]))) * hann
        top = np.sort(resp)[-2:]
        row = [f"frame {t:4d}"]
        for l in (0, L // 2, L - 1):
            row.append(f"L{l}: o {rel(o[f'layer{l}'], tr[f'layer{l}'])[1]:.2e} h {rel(hl[f'layer{l}'], tr[f'layer{l}'])[1]:.2e}")
        row.append(f"feat: o {rel(o['feat'], tr['feat'])[1]:.2e} h {rel(hfeat, tr['feat'])[1]:.2e}")
        row.append(f"head: o {rel(o['head_out'][:, :5], tr['head_out'][:, :5])[0]:.2e} h {rel(hho[:, :5], tr['head_out'][:, :5])[0]:.2e}")
        row.append(f"cell t/o/h {tidx}/{oidx}/{hidx} margin {top[1] - top[0]:.3f}")
        row.append(f"box-truth px: o {np.abs(ob - tb).max():.2f} h {np.abs(hb - tb).max():.2f} o-h {np.abs(ob - hb).max():.2f}"
                   f" | int boxes t {tib.tolist()} o {oib.tolist()} h {hib.tolist()}")
        say("  ".join(row))
        acc["o_last"].append(rel(o[f"layer{L - 1}"], tr[f"layer{L - 1}"])[1])
        acc["h_last"].append(rel(hl[f"layer{L - 1}"], tr[f"layer{L - 1}"])[1])
        acc["oh_last"].append(rel(hl[f"layer{L - 1}"], o[f"layer{L - 1}"])[1])
        acc["o_feat"].append(rel(o["feat"], tr["feat"])[1]); acc["h_feat"].append(rel(hfeat, tr["feat"])[1])
        acc["o_head"].append(rel(o["head_out"][:, :5], tr["head_out"][:, :5])[0])
        acc["h_head"].append(rel(hho[:, :5], tr["head_out"][:, :5])[0])
        acc["o_box"].append(float(np.abs(ob - tb).max())); acc["h_box"].append(float(np.abs(hb - tb).max()))
        acc["oh_box"].append(float(np.abs(ob - hb).max()))
    m = {k: (float(np.mean(v)), float(np.max(v))) for k, v in acc.items()}
    say(f"# summary over {a.frames} frames (mean / max), {time.time() - t0:.0f} s")
    say(f"#   residual stream after the last block, rms rel. to truth: oracle {m['o_last'][0]:.2e} / {m['o_last'][1]:.2e}, "
        f"HIP {m['h_last'][0]:.2e} / {m['h_last'][1]:.2e}; HIP vs oracle {m['oh_last'][0]:.2e} / {m['oh_last'][1]:.2e}")
    say(f"#   feat: oracle {m['o_feat'][0]:.2e} / {m['o_feat'][1]:.2e}, HIP {m['h_feat'][0]:.2e} / {m['h_feat'][1]:.2e}")
    say(f"#   head logits (max rel.): oracle {m['o_head'][0]:.2e} / {m['o_head'][1]:.2e}, HIP {m['h_head'][0]:.2e} / {m['h_head'][1]:.2e}")
    say(f"#   decoded float box vs truth's, px: oracle {m['o_box'][0]:.2f} / {m['o_box'][1]:.2f}, HIP {m['h_box'][0]:.2f} / "
        f"{m['h_box'][1]:.2f}; oracle vs HIP {m['oh_box'][0]:.2f} / {m['oh_box'][1]:.2f}")
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
