"""The evidence chain of bench.py's `roofline` object, on the CPU: the committed rocprofv3 passes under profiles/r05/<tag>/ condense (tools/
roofline_from_profile.py) into exactly the committed profiles/r05/roofline_<tag>.json that bench.py replays (`counters_source`), and the
launch times in it agree with what the bench lines of the same commit report."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles", "r05")


@pytest.mark.parametrize("tag", ["c2", "c1", "c2_queue", "c4", "c4_bvh"])
def test_committed_profile_condenses_to_the_committed_summary(tag):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_from_profile.py"), os.path.join(PROF, tag), tag],
                         check=True, capture_output=True, text=True).stdout
    got, want = json.loads(out), json.load(open(os.path.join(PROF, f"roofline_{tag}.json")))
    assert got == want
    # the two kernel traces agree on the launch time: three in-order steps of the default command, the 20 timed steps of --in-order
    a, b = got["avg_launch_ms_kernel_trace"], got["avg_launch_ms_in_order_run"]
    assert abs(a - b) <= 0.02 * b, (a, b)
    # ... and so do HIP events in the profiled process (bench.py's own figure in that run)
    line = json.loads(open(os.path.join(PROF, tag, "bench_trace_in_order.json")).read().strip().splitlines()[-1])
    assert abs(line["roofline"]["avg_launch_ms"] - b) <= 0.01 * b


def test_headline_bench_line_is_consistent_with_the_profile():
    line = json.loads(open(os.path.join(PROF, "bench_default.json")).read().strip().splitlines()[-1])
    roof = json.load(open(os.path.join(PROF, "roofline_c2.json")))
    r = line["roofline"]
    assert line["metric"].startswith("Msamples/sec") and line["n_gpus"] == 1 and line["config"]["workload"].startswith("C2")
    assert abs(r["avg_launch_ms"] - roof["avg_launch_ms_in_order_run"]) <= 0.02 * r["avg_launch_ms"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["algorithmic_flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) <= 0.01 * r["achieved"]
    assert abs(line["value"] - line["config"]["samples_per_step"] / (line["ms_per_step"] * 1e-3) / 1e6) <= 0.002 * line["value"]
    assert r["traffic"] == pytest.approx(roof["hbm_bytes_per_launch"], rel=0.02)
