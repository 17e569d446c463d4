"""Host logic of PtRenderParams.accel = 1: the BVH builder (pathtrace_amd/csrc/pt_bvh.cpp: binary binned-SAH tree,
collapsed into nodes of up to four children), checked through pt_debug_bvh_check -- no GPU involved.  The entry
rebuilds the tree and verifies, in f64 from the same f32 records the device tests: every object in exactly one leaf
slot with its scan record, every (quantised) child box encloses everything beneath it, 2..4 children per node, the
traversal's stack need as reported and within the stack, padding scale >= scene extent.  depth = deepest leaf of the
binary tree the nodes were collapsed from."""
import numpy as np
import pytest

from test_gpu_fuzz import random_scene


@pytest.mark.parametrize("scene,arg", [(1, 0), (2, 0), (4, 7), (4, 1000), (4, 10000), (4, 150000)])
def test_builtin_scenes(pt, scene, arg):
    objs = pt.builtin_scene(scene, arg)
    depth, nodes, slots = pt.bvh_check(objs)
    assert slots == len(objs)
    assert depth <= 22
    if len(objs) > 4:
        leaves_min = (len(objs) + 3) // 4
        assert nodes >= (leaves_min - 1 + 2) // 3 and nodes < len(objs)      # a node has at most 4 children
        assert depth <= 3 * int(np.ceil(np.log2(len(objs))))        # SAH on these scenes stays near balanced


@pytest.mark.parametrize("n", [0, 1, 2, 4, 5, 8, 9, 63])
def test_tiny_scenes(pt, n):
    rng = np.random.default_rng(n)
    objs = pt.make_objects([(0, list(rng.uniform(-1, 1, 3)) + [0.3], 0, [0.5, 0.5, 0.5]) for _ in range(n)])
    depth, nodes, slots = pt.bvh_check(objs)
    assert slots == n
    if n <= 4:
        assert (depth, nodes) == (0, 0)        # root is the sentinel or one leaf


@pytest.mark.parametrize("seed", range(6))
def test_random_mixed_scenes(pt, seed):
    rng = np.random.default_rng(300 + seed)
    objs = random_scene(pt, rng, int(rng.integers(5, 900)))
    depth, nodes, slots = pt.bvh_check(objs)
    assert slots == len(objs)


def test_coincident_centroids_fall_back_to_median_splits(pt):
    """SAH cannot separate objects with one common centroid; the builder must still terminate with
    <= 4 objects per leaf and a depth the traversal stack can hold."""
    n = 5000
    objs = pt.make_objects([(0, [0.25, -0.5, -2.0, 0.1 + 1e-4 * (i % 7)], 0, [0.5, 0.5, 0.5]) for i in range(n)])
    depth, nodes, slots = pt.bvh_check(objs)
    assert slots == n and depth <= 22
    assert depth == int(np.ceil(np.log2(n / 4)))           # pure object-median tree


def test_clustered_scene_depth_is_bounded(pt):
    """Geometric clusters (each 10x smaller and 10x closer to a corner) push binned SAH towards a degenerate,
    list-like tree; the depth guard switches to median splits before the stack bound."""
    rng = np.random.default_rng(4)
    specs = []
    scale = 1.0
    for level in range(30):
        for _ in range(40):
            c = np.array([1.0, 1.0, -1.0]) * (1.0 - scale) + rng.uniform(-0.4, 0.4, 3) * scale
            specs.append((0, list(c) + [0.01 * scale], 0, [0.5, 0.5, 0.5]))
        scale *= 0.5
    depth, nodes, slots = pt.bvh_check(pt.make_objects(specs))
    assert slots == len(specs) and depth <= 22


def test_non_finite_objects_are_refused(pt):
    """The linear scan's answer for a NaN/inf object depends on the scan order (a NaN t is accepted and then lets
    every later hit through); no tree reproduces that, so accel = 1 refuses the scene instead of guessing."""
    rng = np.random.default_rng(8)
    specs = [(0, list(rng.uniform(-1, 1, 3)) + [0.2], 0, [0.5, 0.5, 0.5]) for _ in range(50)]
    specs[7] = (0, [float("nan"), 0.0, -2.0, 0.3], 0, [0.5, 0.5, 0.5])
    specs[21] = (1, [0, 0, -2, float("inf"), 0, -2, 0, 1, -2], 0, [0.5, 0.5, 0.5])
    specs[30] = (0, [0.0, 0.0, -2.0, float("inf")], 0, [0.5, 0.5, 0.5])
    with pytest.raises(RuntimeError, match="NaN/inf"):
        pt.bvh_check(pt.make_objects(specs))
    assert pt.bvh_check(pt.make_objects(specs[:7]))[2] == 7


def test_bad_arguments(pt):
    objs = pt.make_objects([(7, [0, 0, 0, 1], 0, [0.5, 0.5, 0.5])])
    with pytest.raises(RuntimeError):
        pt.bvh_check(objs)
