"""Builds the in-tree shared libraries with hipcc / g++ (no cmake, no JIT cache):

  gstreamer-vit-tracker_amd/libvittrack_hip.so   HIP kernels + C ABI (include/vittrack_hip.h),
                                                 code object for gfx950 only; exports the boundary's
                                                 symbols and nothing else (version script generated
                                                 from the header)
  gstreamer-vit-tracker_amd/libvittrack_hip_ops.so  the same objects + csrc/vt_ops.hip: the operator-level
                                                 entry points of include/vittrack_hip_ops.h for the
                                                 numerics tests and tuning tools; NOT the product
  harness/libvittrack_host.so                    replay harness: C++ mirror of the reference host
                                                 logic (include/vittrack_host.h); NOT product code,
                                                 libvittrack_hip.so does not link it

hipcc cross-compiles gfx950 without a GPU. The .so files are git-ignored but travel to the GPU
box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "..", "harness")
OBJ = os.path.join(PKG, "build")
LIB_HIP = os.path.join(PKG, "libvittrack_hip.so")
LIB_OPS = os.path.join(PKG, "libvittrack_hip_ops.so")
LIB_HOST = os.path.join(HOST, "libvittrack_host.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_SOURCES = ["k_preproc.hip", "k_gemm.hip", "k_gemm256.hip", "k_attn.hip", "k_misc.hip", "k_head.hip", "k_overlay.hip",
               "vt_engine.hip", "vt_abi.hip", "vt_ingest.hip", "vt_rccl.hip"]
OPS_SOURCES = ["vt_ops.hip"]          # libvittrack_hip_ops.so only
HEADER = os.path.join(PKG, "..", "include", "vittrack_hip.h")


def _write_export_map(path: str) -> None:
    """the product library exports exactly the functions include/vittrack_hip.h declares: the linker's version script is
    generated from the header (tests/test_abi.py compares `nm -D` of the built library with the header)"""
    import re
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", txt)))
    body = "{\n  global:\n" + "".join(f"    {n};\n" for n in names) + "  local:\n    *;\n};\n"
    if not os.path.exists(path) or open(path).read() != body:
        open(path, "w").write(body)
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall",
             "-Wno-unused-function"]
# fused multiply-add allowed where no bit-exact float spec applies (MFMA kernels' epilogues and
# softmax); the pixel stage and the box decode keep one IEEE operation per source operation
FAST_CONTRACT = {"k_gemm.hip", "k_gemm256.hip", "k_attn.hip"}     # not k_head.hip: its decode keeps one IEEE operation per source operation
# k_gemm256.hip: hipcc's SLP vectoriser packs the last FMA of the GELU epilogue into v_pk_fma_f32,
# which has no |x| modifier, so it also emits one v_or per element to build -|x| (and a packed f32
# op issues at the rate of two scalar ones on CDNA4): 8 % more epilogue VALU for nothing
EXTRA_FLAGS = {"k_gemm256.hip": ["-fno-slp-vectorize"]}
HOST_SOURCES = ["host_capi.cpp"]
# translation units whose build identity the library reports (vt_build_info): committed PMC summaries name it, and
# bench.py prints their traffic only for the kernel build they were collected on
STAMPED = {"k_gemm256.hip": ["k_gemm256.hip", "k_gemm_util.hpp", "vt_common.hpp"]}


def tu_sha256(source: str, flags: "list[str]") -> str:
    import hashlib
    h = hashlib.sha256()
    for f in STAMPED[source]:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def build_hip(force: bool = False, save_temps: bool = False, stamps: bool = False, variant: str = "",
              defines: "list[str] | None" = None) -> str:
    """stamps=True: the diagnostic build (-DVT_STAMPS: in-kernel cycle stamps in the GEMM main loops,
    printed by vt_op_gemm_bench) into libvittrack_hip_stamps.so; never loaded by the product - point
    VITTRACK_HIP_LIB at it from a tuning tool."""
    obj_dir, lib_hip, lib_ops = OBJ, LIB_HIP, LIB_OPS
    if stamps:      # diagnostic / tuning builds are ops libraries (boundary + vt_op_*) under their own name
        obj_dir = os.path.join(PKG, "build_stamps")
        lib_hip, lib_ops = None, os.path.join(PKG, "libvittrack_hip_stamps.so")
    if variant:     # tuning builds (python build.py --variant NAME -DMACRO ...): never loaded by the product
        obj_dir = os.path.join(PKG, "build_" + variant)
        lib_hip, lib_ops = None, os.path.join(PKG, f"libvittrack_hip_{variant}.so")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, "vt_common.hpp"), os.path.join(CSRC, "k_gemm_util.hpp"), os.path.join(CSRC, "vt_engine.hpp"),
               os.path.join(PKG, "..", "include", "vittrack_hip.h"), os.path.join(PKG, "..", "include", "vittrack_hip_ops.h")]
    objs, jobs = [], []
    for s in HIP_SOURCES + OPS_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            flags = list(HIP_FLAGS) + (["-DVT_STAMPS"] if stamps else []) + list(defines or [])
            if s in FAST_CONTRACT:
                flags[flags.index("-ffp-contract=off")] = "-ffp-contract=fast"
            flags += EXTRA_FLAGS.get(s, [])
            if s in STAMPED:        # identity of this translation unit: its sources and the flags it is compiled with
                flags.append(f'-DVT_TU_SHA256="{tu_sha256(s, flags)}"')
            cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
            if save_temps:
                cmd += ["-save-temps=obj"]
            jobs.append(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    n_ops = len(OPS_SOURCES)
    if lib_hip:     # the product: the boundary's symbols only
        export_map = os.path.join(obj_dir, "exports.map")
        _write_export_map(export_map)
        if jobs or _newer(lib_hip, [export_map]):
            _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={export_map}", "-o", lib_hip] +
                 objs[:-n_ops])
    if jobs or not os.path.exists(lib_ops):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_ops] + objs)
    return lib_hip or lib_ops


def build_host(force: bool = False) -> str:
    srcs = [os.path.join(HOST, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [
        os.path.join(PKG, "..", "include", "vittrack_host.h"),
        os.path.join(PKG, "..", "include", "vittrack_hip.h")]
    if force or _newer(LIB_HOST, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", LIB_HOST] + srcs +
             ["-ldl", "-lpthread"])
    return LIB_HOST


def build_c_client(force: bool = False) -> str:
    """harness/c_client: the ABI driven from plain C99 (strict flags: the header must be valid C for a
    binding generator); loads libvittrack_hip.so with dlopen at run time"""
    src, out = os.path.join(HOST, "c_client.c"), os.path.join(HOST, "c_client")
    if force or _newer(out, [src, os.path.join(PKG, "..", "include", "vittrack_hip.h")]):
        _run(["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1",
              "-pthread", "-o", out, src, "-ldl"])
    return out


def build_all(force: bool = False):
    return build_hip(force), build_host(force), build_c_client(force)


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--stamps" in sys.argv:
        print(build_hip(force, stamps=True))
        sys.exit(0)
    if "--variant" in sys.argv:
        print(build_hip(force, variant=sys.argv[sys.argv.index("--variant") + 1],
                        defines=[a for a in sys.argv if a.startswith("-D")]))
        sys.exit(0)
    print(build_hip(force, save_temps="--save-temps" in sys.argv))
    if os.path.exists(os.path.join(HOST, "host_capi.cpp")):
        print(build_host(force))
    if os.path.exists(os.path.join(HOST, "c_client.c")):
        print(build_c_client(force))
