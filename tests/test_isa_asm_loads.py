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


def _sreg_range(tok):
    """(first, last) SGPR index of an operand like `s4` / `s[4:5]`, None for anything else"""
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return int(m.group(1)), int(m.group(1))
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    return (int(m.group(1)), int(m.group(2))) if m else None


def _overlap(a, b):
    return a is not None and b is not None and a[0] <= b[1] and b[0] <= a[1]


def _check_function(name, body):
    """forward data flow over the control-flow graph: which VGPRs hold the destination of an inline-asm
    load that no inline-asm `s_waitcnt vmcnt(0)` has retired yet; any instruction touching one is an error.
    Returns the number of asm loads seen.

    Path-sensitive on the constant FLAGS hipcc's control-flow structurizer introduces (round 5): `s_mov_b64 s[a:b], 0 | -1`
    ... `s_and_b64 vcc, exec, s[a:b]` ; `s_cbranch_vccz / vccnz L`. A kernel whose loader waves leave early (k_gemm.hip
    configs 7-10) has its loader code laid out behind the computing path, both guarded by such flags; without following
    them an INFEASIBLE path (flag set to -1, tested as 0) carries the computing waves' pending loads into the loader code.
    A state is (block, known flags); the pending sets of equal states are merged."""
    blocks = _blocks(body)
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab}
    loads = 0
    state_in = {}                              # (block, frozenset(flag items)) -> pending set
    work = []

    def flow(j, pending, flags):
        key = (j, frozenset(flags.items()))
        new = frozenset(pending) if key not in state_in else frozenset(pending | state_in[key])
        if state_in.get(key) != new:
            state_in[key] = new
            work.append(key)

    flow(0, set(), {})
    steps = 0
    while work:
        steps += 1
        assert steps < 200000, f"{name}: flag-sensitive data flow does not converge"
        key = work.pop()
        i, pending, flags = key[0], set(state_in[key]), dict(key[1])
        vcc_const = None                       # True: vcc != 0, False: vcc == 0, None: unknown
        in_asm, fall = False, True
        for line in blocks[i][1]:
            if line.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if line.startswith(";;#ASMEND"):
                in_asm = False
                continue
            toks = line.replace(",", " ").split()
            op = toks[0]
            if in_asm and op.startswith("global_load_dword"):
                pending |= _regs(line.split()[1].rstrip(","))
                loads += 1
                continue
            if in_asm and op == "s_waitcnt" and "vmcnt(0)" in line:
                pending.clear()
                continue
            if op == "s_branch" or op.startswith("s_cbranch"):     # the state AT the branch flows to its target
                target = index[toks[1]]
                if op in ("s_cbranch_vccz", "s_cbranch_vccnz") and vcc_const is not None:
                    taken = vcc_const == (op == "s_cbranch_vccnz")
                    if taken:                  # the only feasible edge
                        flow(target, pending, flags)
                        fall = False
                        break
                    continue                   # never taken: fall through
                flow(target, pending, flags)
                if op == "s_branch":
                    fall = False
                    break
                continue
            if op == "s_endpgm":
                fall = False
                break
            # constant flags and what is derived from them
            # a write that OVERLAPS a tracked pair forgets it: the pair itself, one half of it (s_mov_b32 s4), or a pair that
            # shares a register with it (s[5:6]); a constant move then (re)establishes its own pair
            dst = _sreg_range(toks[1]) if len(toks) > 1 else None
            if dst:
                for k in [k for k in flags if _overlap(_sreg_range(k), dst)]:
                    del flags[k]
            m = re.fullmatch(r"s_mov_b64\s+(s\[\d+:\d+\]),\s*(0|-1)", line.strip())
            if m:
                flags[m.group(1)] = m.group(2) == "-1"
            elif op == "s_and_b64" and len(toks) == 4 and toks[1] == "vcc" and toks[2] == "exec" and toks[3] in flags:
                vcc_const = flags[toks[3]]
            else:
                if len(toks) > 1 and toks[1] == "vcc" or op.startswith("v_cmp") or "vcc" in toks[1:2]:
                    vcc_const = None
                elif op.startswith("v_") and "vcc" in line and not op.startswith("v_cndmask"):
                    vcc_const = None           # carry-out forms write vcc
            hit = _all_vregs(line) & pending
            assert not hit, (f"{name}: `{line}` (block {blocks[i][0]}) touches v{sorted(hit)}, the destination of an "
                             "inline-asm load that has not been waited for")
        if fall and i + 1 < len(blocks):
            flow(i + 1, pending, flags)
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


def test_the_checker_itself_on_synthetic_isa():
    """the data flow catches a touched destination, and follows - only - the constant flags of the structurizer"""
    bad = """
	;;#ASMSTART
	global_load_dwordx4 v[18:21], v[4:5], off
	;;#ASMEND
	v_add_u32_e32 v18, 1, v2
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	s_endpgm
"""
    with pytest.raises(AssertionError, match="touches"):
        _check_function("bad", bad)
    # flag pattern: the load is pending on a path whose flag (-1) makes the branch to the code that reuses v18 infeasible
    flagged = """
	s_cmp_lg_u32 s0, 0
	s_mov_b64 s[2:3], -1
	s_cbranch_scc0 .LBB0_9
	;;#ASMSTART
	global_load_dwordx4 v[18:21], v[4:5], off
	;;#ASMEND
	s_mov_b64 s[4:5], -1
	s_branch .LBB0_5
.LBB0_5:
	s_and_b64 vcc, exec, s[4:5]
	s_cbranch_vccz .LBB0_8
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_add_u32_e32 v18, 1, v18
.LBB0_8:
	s_mov_b64 s[2:3], 0
.LBB0_9:
	s_and_b64 vcc, exec, s[2:3]
	s_cbranch_vccnz .LBB0_11
	s_endpgm
.LBB0_11:
	v_mov_b32_e32 v18, 0
	s_endpgm
"""
    assert _check_function("flagged", flagged) == 1
    # the same with the flags UNKNOWN at their tests (written by compares): the path around the wait into the code that
    # reuses v18 is feasible and must be reported
    unknown = flagged.replace("s_mov_b64 s[4:5], -1", "v_cmp_gt_i32_e64 s[4:5], 1, v2").replace(
        "s_mov_b64 s[2:3], 0", "v_cmp_gt_i32_e64 s[2:3], 1, v3")
    with pytest.raises(AssertionError, match="touches"):
        _check_function("unknown", unknown)
    # a write to ONE HALF of a tracked pair, or to a pair overlapping it, makes the flag unknown as well: the stale constant
    # (-1: the wait is always executed) must not survive it and prune the path around the wait, which - with s[2:3] unknown -
    # reaches the reuse of v18 with the load pending
    for clobber in ("s_mov_b32 s4, s9", "s_mov_b32 s5, 0", "s_or_b64 s[4:5], s[4:5], s[10:11]", "s_mov_b64 s[5:6], 0",
                    "s_and_saveexec_b64 s[4:5], vcc"):
        half = flagged.replace("s_branch .LBB0_5", clobber + "\n\ts_branch .LBB0_5").replace(
            "s_mov_b64 s[2:3], 0", "v_cmp_gt_i32_e64 s[2:3], 1, v3")
        with pytest.raises(AssertionError, match="touches"):
            _check_function("half", half)
    # ... while a write to a DIFFERENT pair leaves it alone
    other = flagged.replace("s_branch .LBB0_5", "s_mov_b32 s6, s9\n\ts_branch .LBB0_5").replace(
        "s_mov_b64 s[2:3], 0", "v_cmp_gt_i32_e64 s[2:3], 1, v3")
    assert _check_function("other", other) == 1
