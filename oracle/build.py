"""Builds oracle/libvt_oracle.so from oracle/vt_oracle.c with gcc (no GPU code).

-ffp-contract=off: no FMA contraction, so each float op is one IEEE binary32 operation and the
HIP kernels (which use __fmul_rn/__fadd_rn) can match bit for bit.

oracle/_ref (a build of the reference's own sources) does not exist for this project: the
reference is a Rust crate and neither cargo nor rustc is in the image, so the
reference is unbuildable here.
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "vt_oracle.c")
LIB = os.path.join(HERE, "libvt_oracle.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    tmp = LIB + f".tmp{os.getpid()}"
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
           "-fopenmp", "-o", tmp, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)   # new inode: a process that has the old library mapped keeps it intact
    return LIB


if __name__ == "__main__":
    print(build(force=True))
