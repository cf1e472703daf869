import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")
    # rebuild stale native libraries (no-op when up to date; hipcc cross-compiles without a GPU)
    import __graft_entry__
    __graft_entry__.build()


def _has_gpu():
    try:
        import gstreamer_vit_tracker_amd as vt
        return os.path.exists(vt.LIB_PATH) and vt.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def vt():
    import gstreamer_vit_tracker_amd as m
    return m


@pytest.fixture(scope="session")
def gpu(vt):
    """The HIP library on a gfx950 device. GPU tests FAIL (not skip) when it is missing: there is
    no fallback path to test instead."""
    assert os.path.exists(vt.LIB_PATH), "libvittrack_hip.so not built (python __graft_entry__.py)"
    n = vt.device_count()
    assert n > 0, "no gfx950 device visible: " + vt.lib().vt_last_error().decode()
    return vt


@pytest.fixture(scope="session")
def oracle():
    from oracle import vit_ref
    vit_ref.lib()
    return vit_ref


@pytest.fixture(scope="session")
def weights_tiny(vt):
    return vt.weights.ensure_weights("tiny")


@pytest.fixture(scope="session")
def weights_cfg2(vt):
    return vt.weights.ensure_weights("cfg2")


@pytest.fixture(scope="session")
def weights_cfg3(vt):
    return vt.weights.ensure_weights("cfg3")


def bf16_round(x):
    from gstreamer_vit_tracker_amd.weights import f32_to_bf16_bits, bf16_bits_to_f32
    return bf16_bits_to_f32(f32_to_bf16_bits(np.asarray(x, np.float32)))


def iou(a, b):
    ax, ay, aw, ah = a
    bx, by, bw, bh = b
    ix = max(0, min(ax + aw, bx + bw) - max(ax, bx))
    iy = max(0, min(ay + ah, by + bh) - max(ay, by))
    inter = ix * iy
    union = aw * ah + bw * bh - inter
    return inter / union if union > 0 else 0.0
