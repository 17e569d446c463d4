"""AddressSanitizer + UndefinedBehaviorSanitizer over everything that runs without a device: scene and camera
constructors, the BVH builder and its verifier (product code), and the oracle.  GPU ASan is not available on the
pool, so the sanitizers run on the CPU build only (tests/tools/run_sanitizers.sh)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.mark.skipif(not os.path.exists(CLANG), reason="no clang++")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "tools", "run_sanitizers.sh"), str(tmp_path / "san_driver")],
                       capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "0 failed checks" in out
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
