"""The C++ host mirror (pathtrace_amd/host/pathtrace.hpp) through examples/cornell:
World::new() -> render() -> export_luminance(), i.e. what the reference's main() does
(src/main.rs:39-67) minus the window."""
import os
import subprocess

import numpy as np
import pytest

from conftest import load_luminance_csv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "cornell")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_example_is_built():
    assert os.path.exists(EXE), "examples/cornell missing: run __graft_entry__.build()"


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present")
def test_cpp_host_fails_loudly_without_gpu(tmp_path):
    r = subprocess.run([EXE, "8", "8", "1", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU path" in r.stderr
    assert not os.path.exists(str(tmp_path / "o_luminance.csv"))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(400, 400), (96, 64)])
def test_cpp_world_render_matches_oracle(pt, orc, tmp_path, w, h):
    """400x400 is World::new() itself; the other size re-authors the scene through the mirrored constructors."""
    spp = 2
    prefix = str(tmp_path / "c")
    r = subprocess.run([EXE, str(w), str(h), str(spp), prefix, "1"], capture_output=True, text=True)   # exact_math
    assert r.returncode == 0, r.stderr
    got = load_luminance_csv(prefix + "_luminance.csv")
    cam = pt.camera_new(width=w, height=h)
    ref, ref8, _ = orc.render(cam, pt.builtin_scene(1), pt.default_params(spp=spp), orc.F32, orc.ITERATIVE, 8)
    assert got.shape == ref.shape
    assert np.abs(got - ref.astype(np.float32)).max() <= 0.5e-6 + 1e-9      # the csv keeps 6 decimals
    with open(prefix + ".ppm", "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == f"{w} {h}\n".encode() and f.readline() == b"255\n"
        rgb = np.frombuffer(f.read(), dtype=np.uint8).reshape(h, w, 3)
    assert np.array_equal(rgb, ref8[..., :3])
