"""The oracle's network against an INDEPENDENT formulation (SURVEY section 8(d):
"cross-checked against PyTorch-CPU"). CPU only.

oracle/vit_ref.py is written in the form the HIP kernels use (split bf16 residual stream, LayerNorm
folded into the consuming GEMM, q pre-scaled to log2 units, p rounded to bf16, the head's 3x3
convolutions as im2col GEMMs) by the same hand as the kernels. oracle/torch_ref.py is the textbook
network in torch.nn.functional, float64: layer_norm, linear, scaled_dot_product_attention, erf gelu,
conv2d. With vit_ref's roundings switched off the two must agree stage by stage to float32 accuracy;
with them on, the bf16 oracle must stay within bf16 noise of the un-quantised answer.

Both follow this build's model specification: the reference's network is unavailable
(/root/reference/Cargo.toml:24, src/main.rs:25) - PARITY UNPINNED; the boundary they stand behind is
VitTrack::init / update (/root/reference/src/tracker_context.rs:88-90)."""
import numpy as np
import pytest


def _patches(vt, oracle, cfg, w, h, sq):
    weights = vt.weights.ensure_weights(cfg)
    sc = vt.synth.MovingSquare(w, h, sq, seed=3)
    trk = oracle.VitTrackRef(weights)
    fr = oracle.Frame.nv12(sc.frame_nv12(4), w, h)
    trk.init(fr, sc.gt_box(4))
    m = trk.m
    srch = trk._pre(fr, trk.box, 4.0, m.S)
    return weights, trk, np.concatenate([trk.tpl, srch], axis=0)


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("cfg,w,h,sq", [("tiny", 640, 480, 64), ("cfg2", 1920, 1080, 64)])
def test_unrounded_oracle_equals_torch_functional(vt, oracle, cfg, w, h, sq):
    from oracle import torch_ref
    weights, trk, patches = _patches(vt, oracle, cfg, w, h, sq)
    truth = torch_ref.TorchModel(weights).forward(patches)
    oracle.ROUNDING = False
    try:
        got = oracle.Model(weights).forward(patches, taps=True)
    finally:
        oracle.ROUNDING = True
    L = trk.m.L
    stages = ["tokens0"] + [f"layer{l}" for l in range(L)] + ["feat", "head_out"]
    worst = {}
    for k in stages:
        a, b = got[k], truth[k]
        if k == "head_out":
            a, b = a[:, :5], b[:, :5]
        worst[k] = _rel(a, b)
    # float32 NumPy against float64 torch: rounding of ~1e3-term sums, nothing structural (measured
    # 1e-7 ... 5e-7 on both models; the bar SURVEY asked for was 1e-4)
    assert max(worst.values()) <= 5e-6, worst


def test_bf16_oracle_stays_within_bf16_noise_of_the_unquantised_answer(vt, oracle):
    from oracle import torch_ref
    weights, trk, patches = _patches(vt, oracle, "cfg2", 1920, 1080, 64)
    truth = torch_ref.TorchModel(weights).forward(patches)
    got = oracle.Model(weights).forward(patches, taps=True)
    L = trk.m.L
    # only the 3-byte pair's half quantum, 2^-13 absolute (numerical specification v3; with the bf16 low half of rounds 3-5: 2^-18 relative)
    assert np.abs(got["tokens0"] - truth["tokens0"]).max() <= 2.0 ** -13 * 1.01
    assert _rel(got[f"layer{L - 1}"], truth[f"layer{L - 1}"]) <= 4e-3      # measured 1.5e-3
    assert _rel(got["feat"], truth["feat"]) <= 8e-3                  # feat itself is rounded to bf16 (measured 3.1e-3)
    # the decision the tracker takes from the map is the same
    hann = trk.m.t["hann"].reshape(-1)
    r_o = 1 / (1 + np.exp(-got["head_out"][:, 0])) * hann
    r_t = 1 / (1 + np.exp(-truth["head_out"][:, 0])) * hann
    assert int(np.argmax(r_o)) == int(np.argmax(r_t))
    assert abs(r_o.max() - r_t.max()) < 0.02


def test_float32_cpu_baseline_computes_the_same_tracker(vt, oracle, weights_cfg2):
    """oracle/cpu_fp32.py - what bench.py's `cpu_baseline` times: float32 torch-CPU network, no bf16 rounding
    emulation - against the bf16 oracle on the same 1080p clip (ViT-B/16 t128/s256), closed loop: same argmax cell,
    identical integer boxes within a pixel, float boxes within 0.1 px (measured 0.01-0.02), scores within 0.01.
    The timed baseline is the same tracker, not a lookalike."""
    from oracle import cpu_fp32
    w, h = 1920, 1080
    sc = vt.synth.MovingSquare(w, h, 64, seed=3)
    a, b = cpu_fp32.VitTrackFp32(weights_cfg2, threads=4), oracle.VitTrackRef(weights_cfg2)
    worst = 0.0
    for t in range(4):
        f = oracle.Frame.nv12(sc.frame_nv12(t), w, h)
        if t == 0:
            a.init(f, sc.gt_box(0))
            b.init(f, sc.gt_box(0))
        ra, rb = a.update(f), b.update(f)
        assert ra.idx == rb.idx and ra.success == rb.success
        assert max(abs(x - y) for x, y in zip(ra.bbox, rb.bbox)) <= 1
        worst = max(worst, float(abs(ra.fbox - rb.fbox).max()))
        assert abs(ra.score - rb.score) < 0.01
    assert worst < 0.1, worst
    assert cpu_fp32.cpu_model_name()
