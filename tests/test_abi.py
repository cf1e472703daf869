"""The C-ABI libraries load and export every symbol their headers declare (no GPU needed; no
compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vth?_[a-z0-9_]+)\s*\(", txt)))


def _exported(path):
    """dynamic symbols a shared library defines (nm -D --defined-only)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_hip_library_exports_every_declared_symbol(vt):
    names = _declared("vittrack_hip.h")
    assert len(names) >= 30
    assert os.path.exists(vt.LIB_PATH), "run python __graft_entry__.py first"
    L = ctypes.CDLL(vt.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(vt.EXPORTS) == names
    assert L.vt_abi_version() == 5


def test_the_product_library_exports_the_boundary_and_nothing_else(vt):
    """`nm -D libvittrack_hip.so` = the functions include/vittrack_hip.h declares: boundary, groups, ingest, converter,
    overlays, RCCL broadcast, diagnostics. No kernel launcher, no engine internals, none of the operator-level test hooks
    (vt_op_*: include/vittrack_hip_ops.h, libvittrack_hip_ops.so)."""
    got = _exported(vt.LIB_PATH)
    assert got == _declared("vittrack_hip.h"), sorted(set(got) ^ set(_declared("vittrack_hip.h")))
    assert not [n for n in got if n.startswith("vt_op_")]
    # the operator header declares only vt_op_* beyond what it includes, and the ops library carries all of them
    ops = [n for n in _declared("vittrack_hip_ops.h") if n.startswith("vt_op_")]
    assert sorted(vt.OPS_EXPORTS) == ops and len(ops) >= 10
    assert os.path.exists(vt.OPS_LIB_PATH), "run python __graft_entry__.py first"
    have = set(_exported(vt.OPS_LIB_PATH))
    assert set(ops) <= have and set(_declared("vittrack_hip.h")) <= have
    # the product path never touches it: no product source includes the ops header, bench.py and the harness call no vt_op_*
    import glob
    pkg = os.path.join(ROOT, "gstreamer-vit-tracker_amd")
    for f in glob.glob(os.path.join(pkg, "csrc", "*")):
        if os.path.basename(f) != "vt_ops.hip":
            assert not re.search(r"#include[^\n]*vittrack_hip_ops", open(f).read()), f
    for f in [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")] + glob.glob(os.path.join(ROOT, "harness", "*")):
        if os.path.isfile(f) and not f.endswith((".so", "c_client")):
            txt = open(f, errors="replace").read()
            assert "ops_lib" not in txt and "vt.op_" not in txt and "vt_op_" not in txt and "vittrack_hip_ops" not in txt, f


def test_host_library_exports_every_declared_symbol(vt):
    from harness import hostlib
    names = _declared("vittrack_host.h")
    L = hostlib.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(hostlib.EXPORTS) == names


def test_struct_layouts_match_header(vt):
    assert ctypes.sizeof(vt.CBBox) == 16
    assert ctypes.sizeof(vt.CResult) == 24       # 24 B of result per frame
    assert ctypes.sizeof(vt.CFrame) == 56       # ABI 2: + window_w, window_h
    assert ctypes.sizeof(vt.CConfig) == 24 + 32
    assert ctypes.sizeof(vt.CKernelTime) == 48 + 8 + 16


def test_no_gpu_means_loud_failure_not_fallback(vt, weights_tiny):
    """Without a gfx950 device creation must fail with VT_ERR_NO_DEVICE; it must never silently
    compute on the CPU."""
    if vt.device_count() > 0:
        return  # on a GPU box the gpu-marked tests cover creation
    import pytest
    with pytest.raises(vt.VtError) as e:
        vt.VitTrack.new(weights_tiny)
    assert e.value.code == -2
    with pytest.raises(vt.VtError):
        vt.nv12_full_to_rgb(__import__("numpy").zeros(96, "uint8"), 8, 8)


def test_header_is_strict_c99_and_gives_a_c_compiler_the_same_layouts(vt):
    """harness/c_client.c includes include/vittrack_hip.h from plain C (what bindgen / cgo would
    consume) and is compiled with -std=c99 -Wall -Wextra -Werror -pedantic by build.py; the struct
    sizes a C compiler derives from the header must be the ones the ctypes binding uses"""
    import ctypes
    import subprocess
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_vt_build", os.path.join(root, "gstreamer-vit-tracker_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    exe = b.build_c_client()
    out = subprocess.run([exe, "sizes"], capture_output=True, text=True, check=True).stdout.split()
    got = dict(zip(out[0::2], (int(x) for x in out[1::2])))
    assert got["vt_config"] == ctypes.sizeof(vt.CConfig)
    assert got["vt_model_info"] == ctypes.sizeof(vt.CModelInfo)
    assert got["vt_frame"] == ctypes.sizeof(vt.CFrame)
    assert got["vt_result"] == ctypes.sizeof(vt.CResult) and got["vt_bbox"] == 16
    assert got["abi"] == vt.lib().vt_abi_version() and got["max_streams"] == 1024


def test_every_tool_script_compiles():
    """tools/*.py are run by hand on GPU boxes: a syntax error there costs a box acquisition; shell tools get `bash -n`"""
    import glob
    import py_compile
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "tools", "*.py"))):
        py_compile.compile(f, doraise=True)
    for f in sorted(glob.glob(os.path.join(root, "tools", "*.sh"))):
        assert subprocess.run(["bash", "-n", f], capture_output=True).returncode == 0, f
