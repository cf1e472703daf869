"""bindings/vit_tracker (the Rust crate a maintainer drops in for the reference's `../vit_tracker`,
/root/reference/Cargo.toml:24) is pinned to include/vittrack_hip.h: cargo/rustc do not exist in the
build image, so instead of compiling it this test PARSES both sides and fails on drift -
every function of the header declared in src/sys.rs with the same argument count, order and
types; every #[repr(C)] struct with the same fields, order and size as the C struct (sizes also
checked against what the ctypes binding and a C99 compiler derive: tests/test_abi.py). The safe
wrapper (src/lib.rs) must keep the names the reference's call sites use."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "bindings", "vit_tracker")

C_SCALARS = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "float": "f32", "double": "f64",
             "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "char": "c_char", "uint8_t": "u8",
             "uint16_t": "u16", "void": "void"}
RUST_SCALARS = {"c_int": "i32", "i32": "i32", "u32": "u32", "f32": "f32", "f64": "f64", "u64": "u64", "i64": "i64",
                "usize": "usize", "c_char": "c_char", "u8": "u8", "u16": "u16", "c_void": "void"}
SIZES = {"i32": 4, "u32": 4, "f32": 4, "f64": 8, "u64": 8, "i64": 8, "usize": 8, "c_char": 1, "u8": 1, "u16": 2, "ptr": 8}
STRUCT_ALIAS = {"BBox": "vt_bbox"}          # the reference's own name for it (src/selection_state.rs:44)


def _snake(name):
    return STRUCT_ALIAS.get(name) or re.sub(r"(?<!^)(?=[A-Z])", "_", name).lower()


def _strip_c(txt):
    return re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


def _c_type(t, structs):
    """canonical form of a C type string (no name): 'ptr', a scalar, or a struct name"""
    t = t.replace("const", " ").replace("struct", " ").strip()
    if "*" in t:
        return "ptr"
    t = t.split()[-1] if t.split() else t
    if t in C_SCALARS:
        return C_SCALARS[t]
    if t in ("vt_status", "vt_pixfmt", "vt_draw_type"):
        return "i32"
    assert t in structs, f"unknown C type {t!r}"
    return t


def parse_header():
    txt = _strip_c(open(os.path.join(ROOT, "include", "vittrack_hip.h")).read())
    txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", txt, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "int32_t a, b", "const void* plane0", "char text[36]", "int32_t reserved[6]"
            mm = re.match(r"(.*?[\s\*])(\w+(?:\[\d+\])?(?:\s*,\s*\w+(?:\[\d+\])?)*)$", decl)
            assert mm, decl
            base = mm.group(1)
            for nm in mm.group(2).split(","):
                nm = nm.strip()
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                ty = _c_type(base, structs)
                fields.append((arr.group(1), ty, int(arr.group(2))) if arr else (nm, ty, 0))
        structs[name] = fields
    opaque = set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", txt))
    funcs = {}
    body = re.sub(r"typedef\s+(?:struct|enum)\s+\w+\s*\{.*?\}\s*\w+\s*;", "", txt, flags=re.S)
    for m in re.finditer(r"([\w\s\*]+?)\b(vt_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", body, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if re.search(r"\[\w*\]$", a):                # `const uint8_t id[VT_RCCL_ID_BYTES]` decays to a pointer
                    params.append("ptr")
                    continue
                mm = re.match(r"(.*?[\s\*])(\w+)$", a)
                assert mm, a
                params.append(_c_type(mm.group(1), structs | {o: [] for o in opaque}))
        rt = _c_type(ret, structs | {o: [] for o in opaque})
        funcs[name] = (rt, params)
    return structs, funcs


def _rust_type(t):
    t = t.strip()
    if t.startswith("*const") or t.startswith("*mut"):
        return "ptr"
    arr = re.match(r"\[(\w+)\s*;\s*(\d+)\]", t)
    if arr:
        return (RUST_SCALARS[arr.group(1)], int(arr.group(2)))
    if t in RUST_SCALARS:
        return RUST_SCALARS[t]
    return _snake(t)


def parse_sys_rs():
    txt = open(os.path.join(CRATE, "src", "sys.rs")).read()
    txt = re.sub(r"//.*$", "", txt, flags=re.M)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\)]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\}", txt, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields = []
        for f in re.finditer(r"(?:pub\s+)?(r#)?(\w+)\s*:\s*([^,\n]+),", body):
            ty = _rust_type(f.group(3))
            fields.append((f.group(2), ty[0], ty[1]) if isinstance(ty, tuple) else (f.group(2), ty, 0))
        structs[name] = fields
    ext = re.search(r'extern "C" \{(.*)\}', txt, flags=re.S).group(1)
    funcs = {}
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", ext, flags=re.S):
        params = [_rust_type(p.split(":", 1)[1]) for p in m.group(2).split(",") if ":" in p]
        funcs[m.group(1)] = (_rust_type(m.group(3)) if m.group(3) else "void", params)
    consts = dict(re.findall(r"pub const (\w+): \w+ = (-?\d+);", txt))
    return structs, funcs, consts


def _size(fields, all_structs):
    """C layout size of a struct of naturally aligned scalars (what #[repr(C)] and the C compiler both do)"""
    off, align = 0, 1
    for _, ty, n in fields:
        if ty in SIZES:
            sz = al = SIZES[ty]
        else:
            sz, al = _size(all_structs[ty], all_structs), 4
        off = (off + al - 1) // al * al
        off += sz * max(n, 1)
        align = max(align, al)
    return (off + align - 1) // align * align


def test_every_header_function_is_declared_identically_in_sys_rs():
    _, cf = parse_header()
    _, rf, _ = parse_sys_rs()
    assert len(cf) >= 55 and not [n for n in cf if n.startswith("vt_op_")]
    assert sorted(rf) == sorted(cf), (sorted(set(cf) - set(rf)), sorted(set(rf) - set(cf)))
    for name, (ret, params) in cf.items():
        rret, rparams = rf[name]
        assert rret == ret, (name, ret, rret)
        assert rparams == params, (name, params, rparams)


def test_repr_c_structs_match_the_header_field_for_field(vt):
    cs, _ = parse_header()
    rs, _, consts = parse_sys_rs()
    rust_by_c = {_snake(k): v for k, v in rs.items() if v}          # opaque handles have no fields
    for cname in ("vt_bbox", "vt_result", "vt_config", "vt_model_info", "vt_frame", "vt_draw_cmd", "vt_kernel_time"):
        assert cname in rust_by_c, f"{cname} not bound in sys.rs"
        assert rust_by_c[cname] == cs[cname], (cname, cs[cname], rust_by_c[cname])
    # sizes: the same numbers the ctypes binding and a C99 compiler give (tests/test_abi.py)
    want = {"vt_bbox": vt.CBBox, "vt_result": vt.CResult, "vt_config": vt.CConfig, "vt_model_info": vt.CModelInfo,
            "vt_frame": vt.CFrame, "vt_draw_cmd": vt.CDrawCmd, "vt_kernel_time": vt.CKernelTime}
    for cname, ct in want.items():
        assert _size(rust_by_c[cname], cs) == ctypes.sizeof(ct), cname
    # vt_config names the three fields that were once `reserved` slots
    names = [f[0] for f in rust_by_c["vt_config"]]
    assert names[6:9] == ["max_device_mib", "host_window_margin_pct", "host_zero_copy"]
    assert rust_by_c["vt_config"][-1] == ("reserved", "i32", 5)
    hdr = open(os.path.join(ROOT, "include", "vittrack_hip.h")).read()
    assert int(consts["VT_ABI_VERSION"]) == int(re.search(r"#define VT_ABI_VERSION (\d+)", hdr).group(1))
    assert int(consts["VT_MAX_STREAMS"]) == int(re.search(r"#define VT_MAX_STREAMS (\d+)", hdr).group(1))
    for k, v in re.findall(r"(VT_(?:OK|ERR_\w+)) = (-?\d+)", hdr):
        assert int(consts[k]) == int(v), k


def test_safe_wrapper_keeps_the_reference_call_surface():
    """names and shapes the reference's call sites need (src/tracker_context.rs:2,21,88,90,94,120,123;
    src/selection_state.rs:1,44)"""
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    for needle in ("pub struct VitTrack", "unsafe impl Send for VitTrack", "pub fn new(model_path: &str) -> Result<Self, TrackError>",
                   "pub fn init(&mut self, img: &ArrayView3<u8>, bbox: BBox)",
                   "pub fn update(&mut self, img: &ArrayView3<u8>) -> Result<TrackResult, TrackError>",
                   "pub fn new(x: i32, y: i32, width: i32, height: i32) -> Self", "pub fn from_array(a: &[i32; 4]) -> Self",
                   "pub success: bool", "pub score: f32", "pub bbox: [i32; 4]", "impl Drop for VitTrack"):
        assert needle in lib, needle
    # the host is built with panic = "abort" (/root/reference/Cargo.toml:37): nothing in the wrapper may panic. No
    # rustc here to prove it, so the constructs that can are refused outright (comments stripped first)
    code = re.sub(r"//.*$", "", lib, flags=re.M)
    for banned in ("assert!", "assert_eq!", "debug_assert!", ".unwrap()", ".expect(", "panic!", "unreachable!", "todo!", "unimplemented!"):
        assert banned not in code, f"{banned} in bindings/vit_tracker/src/lib.rs"
    assert not re.search(r"\[[^\]\n]*\.\.[^\]\n]*\]", code), "range indexing of a slice can panic: use .get(..)"
    # a view the library cannot take is an Err of update, and init's failure is kept for the next update
    assert "fn rgb_view(img: &ArrayView3<u8>) -> Result<" in lib and "self.pending.take()" in lib
    toml = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert re.search(r'^name = "vit_tracker"$', toml, flags=re.M) and re.search(r'^version = "0.1.0"$', toml, flags=re.M)
    # every sys:: function the wrapper calls exists in sys.rs
    _, rf, _ = parse_sys_rs()
    used = set(re.findall(r"sys::(vt_\w+)\(", lib))
    assert used and used <= set(rf), used - set(rf)
    # INTEGRATION.md points at the files instead of restating them
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "bindings/vit_tracker/src/lib.rs" in integ and "fn vt_update_rgb8(" not in integ
