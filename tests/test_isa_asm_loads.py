"""Build-time check of the hand-waited inline-asm loads (k_gemm.hip's early addend loads).

The 4-wave GEMM kernel fetches its epilogue inputs (old residual pair / positional rows, bias, the folded
LayerNorm's chunk partials) with inline-asm `global_load_dwordx4` BEFORE the main loop and waits for them
with a hand-written `s_waitcnt vmcnt(0)` after it. hipcc's waitcnt pass does not see loads inside inline
asm: if register pressure ever made it copy, spill or reuse one of those destination registers between
the load and the wait, the kernel would read stale registers with no diagnostic. This test compiles
k_gemm.hip for gfx950 with -save-temps (hipcc cross-compiles without a GPU) and checks in the ISA of
every kernel that has such loads: no instruction touches a destination register of an asm load between
the load and the asm wait that names it, and those kernels use no scratch memory. CPU only."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gstreamer-vit-tracker_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _regs(tok):
    """'v[12:15]' -> {12..15}, 'v7' -> {7}"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def _all_vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", line):
        out |= _regs(tok)
    return out


def _isa_of(source):
    """ISA text of one source file, compiled with exactly the flags build.py gives it"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vt_build", os.path.join(ROOT, "gstreamer-vit-tracker_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    flags = list(b.HIP_FLAGS)
    if source in b.FAST_CONTRACT:
        flags[flags.index("-ffp-contract=off")] = "-ffp-contract=fast"
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [HIPCC] + flags + b.EXTRA_FLAGS.get(source, []) + ["-c", os.path.join(CSRC, source),
                                                                 "-o", os.path.join(tmp, "k.o"), "-save-temps=obj"]
        subprocess.run(cmd, check=True, capture_output=True, cwd=tmp)
        path = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
        return open(path).read()


def _blocks(body):
    """basic blocks of one function's text: [(label or None, [instruction lines])] in layout order"""
    blocks, cur, label = [], [], None
    for raw in body.splitlines():
        line = raw.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            blocks.append((label, cur))
            cur, label = [], m.group(1)
            continue
        if not line or line.startswith(";") and not line.startswith(";;#ASM") or line.startswith("."):
            continue
        cur.append(line)
    blocks.append((label, cur))
    return blocks


def _check_function(name, body):
    """forward data flow over the control-flow graph: which VGPRs hold the destination of an inline-asm
    load that no inline-asm `s_waitcnt vmcnt(0)` has retired yet; any instruction touching one is an error.
    Returns the number of asm loads seen."""
    blocks = _blocks(body)
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab}
    loads = 0
    state_in = [None] * len(blocks)
    state_in[0] = frozenset()
    work = [0]

    def flow(j, pending):
        new = frozenset(pending) if state_in[j] is None else frozenset(pending | state_in[j])
        if new != state_in[j]:
            state_in[j] = new
            work.append(j)

    while work:
        i = work.pop()
        pending = set(state_in[i])
        in_asm, fall = False, True
        for line in blocks[i][1]:
            if line.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if line.startswith(";;#ASMEND"):
                in_asm = False
                continue
            op = line.split()[0]
            if in_asm and op.startswith("global_load_dword"):
                pending |= _regs(line.split()[1].rstrip(","))
                loads += 1
                continue
            if in_asm and op == "s_waitcnt" and "vmcnt(0)" in line:
                pending.clear()
                continue
            if op == "s_branch" or op.startswith("s_cbranch"):     # the state AT the branch flows to its target
                flow(index[line.split()[1]], pending)
                if op == "s_branch":
                    fall = False
                    break
                continue
            if op == "s_endpgm":
                fall = False
                break
            hit = _all_vregs(line) & pending
            assert not hit, (f"{name}: `{line}` (block {blocks[i][0]}) touches v{sorted(hit)}, the destination of an "
                             "inline-asm load that has not been waited for")
        if fall and i + 1 < len(blocks):
            flow(i + 1, pending)
    return loads


# (source, kernel name pattern, at least this many kernels carry hand-waited loads)
@pytest.mark.parametrize("source,pattern,at_least", [
    ("k_gemm.hip", r"gemm_bf16_kernel", 10),     # 64x64 X-epilogues + every bf16 epilogue with a folded LayerNorm
    ("k_gemm256.hip", r"gemm256p?_kernel", 2),   # the last-arriver finalize of the X-epilogues' row terms
    ("k_head.hip", r"head_conv_kernel", 14),     # fused tail: the stream state prefetched by thread 0, sc1 hand-off loads
])
def test_asm_load_destinations_are_untouched_until_their_wait(source, pattern, at_least):
    isa = _isa_of(source)
    kernels = re.findall(r"^(_Z\w*" + pattern + r"\w+):[^\n]*\n(.*?)\n\.Lfunc_end", isa, re.S | re.M)
    assert kernels
    checked = 0
    for name, body in kernels:
        if _check_function(name, body):
            checked += 1
            meta = re.search(r"\.name:\s+" + re.escape(name) + r"\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", isa)
            assert meta and int(meta.group(1)) == 0, f"{name}: uses scratch memory beside hand-waited loads"
    assert checked >= at_least, checked
