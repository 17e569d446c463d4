"""The C-ABI library: loads, exports every symbol include/pathtrace_amd.h declares,
struct layouts agree with the header, host-side helpers (camera, scenes, tile
partition) behave like the reference, and -- without a GPU -- rendering fails
loudly instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_header_symbols_are_all_exported_and_bound(pt):
    hdr = open(os.path.join(ROOT, "include", "pathtrace_amd.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = pt._lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(pt._lib.SYMBOLS), declared ^ set(pt._lib.SYMBOLS)
    assert lib.pt_abi_version() == int(re.search(r"#define PT_ABI_VERSION (\d+)", hdr).group(1))


def test_struct_layouts_match_the_c_compiler(pt, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pathtrace_amd.h"\n'
                   'int main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(PtCamera), sizeof(PtObject),'
                   ' sizeof(PtRenderParams), sizeof(PtStats), offsetof(PtRenderParams, t_min),'
                   ' offsetof(PtRenderParams, max_paths_in_flight), offsetof(PtStats, bounce_kernel_ms));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    L = pt._lib
    assert got == [C.sizeof(L.PtCamera), C.sizeof(L.PtObject), C.sizeof(L.PtRenderParams), C.sizeof(L.PtStats),
                   L.PtRenderParams.t_min.offset, L.PtRenderParams.max_paths_in_flight.offset,
                   L.PtStats.bounce_kernel_ms.offset]


def test_default_params_are_the_reference_constants(pt):
    p = pt.default_params()
    assert (p.spp, p.min_depth, p.max_depth, p.integrator, p.t_min) == (3000, 4, 50, 0, 0.001)   # world.rs:18, rendering.rs:6-7
    assert (p.band_index, p.band_count, p.spp_offset) == (0, 1, 0)
    assert (p.exact_math, p.accel) == (0, 2)          # fast arithmetic; PT_ACCEL_AUTO (same film either way)


def test_builtin_scene_1_is_world_new(pt):
    """src/world.rs:80-211: 12 triangles (10 wall + 2 light) then the glass sphere; lights = [10, 11]."""
    objs = pt.builtin_scene(1)
    assert len(objs) == 13
    assert [o.shape_tag for o in objs] == [1] * 12 + [0]
    assert [o.mat_tag for o in objs] == [0] * 10 + [1, 1] + [2]
    alb = [tuple(o.mat[0:3]) for o in objs[:10]]
    assert alb == [(0.8, 0.1, 0.1)] * 2 + [(0.1, 0.8, 0.1)] * 2 + [(0.2, 0.2, 0.8)] * 2 + [(0.2, 0.8, 0.8)] * 2 + \
        [(0.8, 0.8, 0.8)] * 2
    for o in objs[:10]:                       # every wall vertex is a corner of [-1,1]^2 x [-3,-1]
        v = np.array(o.shape[:9]).reshape(3, 3)
        assert np.all(np.abs(v[:, :2]) == 1.0) and np.all(np.isin(v[:, 2], [-3.0, -1.0]))
    # wall planes: left x=-1, right x=+1, back z=-3, floor y=-1, ceiling y=+1
    for k, (axis, val) in enumerate([(0, -1), (0, -1), (0, 1), (0, 1), (2, -3), (2, -3), (1, -1), (1, -1), (1, 1), (1, 1)]):
        assert np.all(np.array(objs[k].shape[:9]).reshape(3, 3)[:, axis] == val)
    for o in objs[10:12]:
        v = np.array(o.shape[:9]).reshape(3, 3)
        assert np.all(v[:, 1] == 1.0 - 0.01) and np.all(np.abs(v[:, 0]) == 0.3)
        assert set(np.round(v[:, 2], 12)) <= {-2.3, -1.7} and tuple(o.mat[:3]) == (15.0, 15.0, 15.0)
    # the two light triangles tile the square: total area 0.36
    area = sum(0.5 * np.linalg.norm(np.cross(*(np.array(o.shape[:9]).reshape(3, 3)[1:] - np.array(o.shape[:3]))))
               for o in objs[10:12])
    assert area == pytest.approx(0.36)
    g = objs[12]
    assert tuple(g.shape[:4]) == (0.4, -0.6, -2.0, 0.4) and tuple(g.mat[:6]) == (0.3, 1.0, 1.0, 1.0, 0.0, 1.5)


def test_builtin_scene_2_and_4(pt):
    c2 = pt.builtin_scene(2)
    assert len(c2) == 10 and all(o.shape_tag == 0 for o in c2)
    assert [o.mat_tag for o in c2] == [0] * 5 + [1] + [0] * 4
    assert tuple(c2[5].shape[:4]) == (0.0, 1.0 - 0.21, -2.0, 0.2) and tuple(c2[5].mat[:3]) == (36.0,) * 3
    for o, (axis, sign) in zip(c2[:5], [(0, -1), (0, 1), (2, -1), (1, -1), (1, 1)]):
        c, r = np.array(o.shape[:3]), o.shape[3]
        assert r == 100.0
        # tangent to the reference's wall plane, centred on the box axis
        plane = {0: 1.0, 1: 1.0, 2: 1.0}[axis]
        centre_off = c[axis] - (-2.0 if axis == 2 else 0.0)
        assert centre_off == sign * (plane + r)
    c4 = pt.builtin_scene(4, 1000)
    assert len(c4) == 1000
    ctr = np.array([o.shape[:3] for o in c4]); rad = np.array([o.shape[3] for o in c4])
    assert np.all(np.abs(ctr[:, :2]) < 1) and np.all((ctr[:, 2] > -3) & (ctr[:, 2] < -1))
    assert np.all((rad > 0.005) & (rad < 0.03))
    assert [i for i, o in enumerate(c4) if o.mat_tag == 1] == list(range(0, 1000, 100))
    # a prefix of a larger scene is the smaller scene (sphere i depends on i only)
    c4b = pt.builtin_scene(4, 50)
    assert all(tuple(a.shape) == tuple(b.shape) and tuple(a.mat) == tuple(b.mat) for a, b in zip(c4b, c4))
    assert len(pt.builtin_scene(4)) == 10000


def test_tile_partition(pt):
    H = 37
    for band_rows, G in [(1, 1), (4, 3), (5, 8), (0, 1), (64, 2)]:
        seen = []
        for g in range(G):
            rows = pt.tile_row_indices(H, band_rows, g, G)
            assert pt.tile_rows(H, band_rows, g, G) == len(rows)
            assert rows == sorted(rows)
            seen += rows
        assert sorted(seen) == list(range(H))          # disjoint cover of the image
    assert pt.tile_rows(8, 64, 1, 2) == 0              # a rank may own nothing (ragged partition)


def test_camera_rejects_bad_arguments(pt):
    with pytest.raises(pt._lib.PtError):
        pt.camera_new(width=0, height=10)
    n = C.c_uint32(0)
    assert pt._lib.lib().pt_builtin_scene(99, 0, None, 0, C.byref(n)) != 0


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present; the no-device error path cannot be exercised")
def test_no_gpu_means_loud_failure_not_fallback(pt):
    with pytest.raises(pt._lib.PtError) as e:
        pt.Context(0)
    assert e.value.code == 2 and "no CPU path" in str(e.value)      # PT_ERR_NO_DEVICE
    cam = pt.camera_new(width=8, height=8)
    with pytest.raises(pt._lib.PtError):
        pt.render_host(cam, pt.builtin_scene(2), pt.default_params(spp=1))


def test_product_does_not_reference_the_oracle():
    """The product path may not import, include or link anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "pathtrace_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"oracle/|from oracle|import oracle|liborc|pt_oracle", txt):
                    bad.append(os.path.join(base, f))
    assert not bad, bad
    needed = subprocess.check_output(["readelf", "-d", os.path.join(ROOT, "pathtrace_amd", "libpathtrace_amd.so")],
                                     text=True)
    assert "liborc" not in needed and "amdhip64" in needed
