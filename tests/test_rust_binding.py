"""The Rust -sys crate (rust/pathtrace-amd-sys) is source only here -- no cargo/rustc in the image.
What can be checked without a compiler is that it declares exactly the header's ABI: every function with the
same argument count, every struct with the same fields in the same order and the same scalar types."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "pathtrace_amd.h")).read()
RUST = open(os.path.join(ROOT, "rust", "pathtrace-amd-sys", "src", "lib.rs")).read()

C2RUST = {"double": "f64", "float": "f32", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "uint8_t": "u8", "int": "c_int"}


def _strip_comments(src):
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _c_functions():
    out = {}
    for m in re.finditer(r"^(?:const\s+)?[a-z_0-9A-Z]+\*?\s+\**(pt_[a-z_]+)\(([^;]*?)\);", _strip_comments(HEADER), flags=re.M | re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return out


def _rust_functions():
    out = {}
    for m in re.finditer(r"pub fn (pt_[a-z_]+)\((.*?)\)\s*(?:->[^;]+)?;", RUST, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    return out


def test_every_header_function_is_declared_with_the_same_arity():
    c, r = _c_functions(), _rust_functions()
    assert len(c) >= 39 and "pt_render_pixels" in c and "pt_multi_render_device" in c and "pt_render_device" in c and "pt_debug_bvh_check" in c
    assert c == r


def _c_struct(name):
    body = re.search(r"typedef struct \{([^}]*)\} %s;" % name, _strip_comments(HEADER)).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(\w+)\s+(.*)", decl)
        for var in m.group(2).split(","):
            var = var.strip()
            am = re.match(r"(\w+)\[(\d+)\]", var)
            fields.append((am.group(1), "[%s; %s]" % (C2RUST[m.group(1)], am.group(2))) if am else (var, C2RUST[m.group(1)]))
    return fields


def _rust_struct(name):
    body = re.search(r"pub struct %s \{(.*?)\}" % name, RUST, flags=re.S).group(1)
    return [(m.group(1), m.group(2).strip()) for m in re.finditer(r"pub (\w+): ([^,]+),", body)]


def test_struct_fields_match_the_header():
    for name in ("PtCamera", "PtObject", "PtRenderParams", "PtTuning", "PtStats"):
        assert _c_struct(name) == _rust_struct(name), name


def test_constants_match_the_header():
    hdr = _strip_comments(HEADER)
    for m in re.finditer(r"(PT_(?:OK|ERR|SHAPE|MAT|INTEGRATOR|ACCEL)_?\w*)\s*=\s*(\d+)", hdr):
        assert re.search(r"pub const %s: \w+ = %s;" % (m.group(1), m.group(2)), RUST), m.group(1)
    assert re.search(r"#define PT_ABI_VERSION (\d+)", HEADER).group(1) == re.search(r"PT_ABI_VERSION: u32 = (\d+);", RUST).group(1)


def test_reference_patch_applies(tmp_path):
    """rust/reference-shim/reference-gpu.patch (describe() on the traits, Camera::to_pod, gpu::render in main) must
    apply to the reference's sources.  Only where the reference is present (this container; not the GPU box)."""
    import shutil
    import subprocess
    import pytest
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "src")):
        pytest.skip("reference sources not present")
    patch = os.path.join(ROOT, "rust", "reference-shim", "reference-gpu.patch")
    touched = sorted(set(re.findall(r"^\+\+\+ b/(\S+)", open(patch).read(), flags=re.M)))
    assert "src/main.rs" in touched and "src/objects/shape.rs" in touched and "src/objects/material.rs" in touched
    for f in touched:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(ref, f), tmp_path / f)
    r = subprocess.run(["patch", "-p1", "--dry-run", "-i", patch], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    # every impl of the two traits in the reference gets a description (or the trait's default)
    shape_impls = len(re.findall(r"^impl Shape for", open(os.path.join(ref, "src/objects/shape.rs")).read(), flags=re.M))
    assert open(patch).read().count("fn describe(&self) -> (u32, [f64; 9])") == shape_impls + 1        # + the trait itself
    gpu_rs = open(os.path.join(ROOT, "rust", "reference-shim", "gpu.rs")).read()
    assert "unsafe" not in gpu_rs.replace("forbid(unsafe_code)", "").replace("`unsafe`", "")
