"""bench.py's launch logic (no GPU): `python bench.py --gpus N` must work however it is started -- as a plain process it
takes the library's single-process multi-device path (pt_multi_*), under torch.distributed.run the one-process-per-GPU
path -- and never exits just because no launcher environment is present."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_plain_process_launches(bench):
    assert bench.launch_mode(1, {}) == ("single", 1, 0, 0)
    for n in (2, 4, 8):
        assert bench.launch_mode(n, {}) == ("multi", n, 0, 0)              # the driver's `python3 bench.py --gpus N` form
    assert bench.launch_mode(1, {}, force_multi=True) == ("multi", 1, 0, 0)
    assert bench.launch_mode(1, {}, force_dist=True) == ("dist", 1, 0, 0)
    assert bench.launch_mode(1, {"WORLD_SIZE": "1", "RANK": "0"}) == ("single", 1, 0, 0)


def test_torchrun_environment_wins_over_the_flag(bench):
    env = {"WORLD_SIZE": "8", "RANK": "5", "LOCAL_RANK": "5"}
    assert bench.launch_mode(8, env) == ("dist", 8, 5, 5)
    assert bench.launch_mode(1, env) == ("dist", 8, 5, 5)
    with pytest.raises(SystemExit):
        bench.launch_mode(8, env, force_multi=True)
    with pytest.raises(SystemExit):
        bench.launch_mode(1, {}, force_dist=True, force_multi=True)


def test_workloads_name_the_baseline_configs(bench):
    assert bench.WORKLOADS["c2"][2:5] == (1024, 1024, 64)                   # BASELINE.json configs[1]: the headline
    assert bench.WORKLOADS["c3"][4] == 4096 and bench.WORKLOADS["c4"][1] == 10000 and bench.WORKLOADS["c5"][2:5] == (3840, 2160, 1024)


def test_cpu_baseline_leg_on_a_tiny_workload(bench, pt):
    """The bounded CPU sample: figures for the 16-thread share, for every visible CPU, and BASELINE configs[0]."""
    out = bench.cpu_baseline(pt, pt.builtin_scene(2), 64, 64, 2, "C2 scene")
    assert out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1
    assert out["all_cores"]["cores"] == len(os.sched_getaffinity(0)) and out["all_cores"]["value"] > 0
    assert out["config0_single_thread"]["cores"] == 1


def test_the_timed_regions_counters_are_checked_against_the_job(bench):
    """bench.py's own verification of its overlapped timed region: exact device counters, or a problem."""
    per3 = {"vertices": 3 * 329_000_001, "shadow_rays": 3 * 250_000_007}
    good = {"samples": 20 * 67_108_864, "samples_expected": 20 * 67_108_864, "vertices": 20 * 329_000_001, "shadow_rays": 20 * 250_000_007}
    assert bench.counter_problems("t", good, 20 * 67_108_864, 20, per3, 3) == []
    assert bench.counter_problems("t", good, 20 * 67_108_864, 20) == []
    lost = dict(good, samples=int(good["samples"] * 0.97))                      # round 4's race: 3 % of the samples lost
    assert any("finished" in p for p in bench.counter_problems("t", lost, 20 * 67_108_864, 20, per3, 3))
    off = dict(good, vertices=good["vertices"] - 64)
    assert any("vertices" in p for p in bench.counter_problems("t", off, 20 * 67_108_864, 20, per3, 3))
    assert any("vertices" in p for p in bench.counter_problems("t", dict(good, vertices=good["vertices"] + 1), 20 * 67_108_864, 20))
    wrong_job = bench.counter_problems("t", good, 19 * 67_108_864, 20, per3, 3)
    assert wrong_job and "renders of the job" in wrong_job[0]
