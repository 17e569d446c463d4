"""N>1 path on CPU: world_size-2/3/4 gloo processes each produce their interleaved row
bands (the oracle stands in for the GPU renderer here) and rank 0 gathers the
frame with pathtrace_amd.dist.gather_tiles -- the same code path bench.py runs
over RCCL.  The assembled frame must equal the single-process render bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, band_rows, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pathtrace_amd as pt
    from pathtrace_amd.dist import gather_film, gather_tiles
    from oracle import orc
    cam = pt.camera_new(width=W, height=H)
    prm = pt.default_params(spp=2, band_rows=band_rows, band_index=rank, band_count=world)
    lin, rgba, _ = orc.render(cam, pt.builtin_scene(1), prm, orc.F32, orc.ITERATIVE)
    frame = gather_tiles(torch.from_numpy(lin.astype(np.float32)), H, band_rows, rank, world)
    frame8 = gather_tiles(torch.from_numpy(rgba), H, band_rows, rank, world)
    # the packed single-collective form bench.py uses must give the same two planes
    pf, p8 = gather_film(torch.from_numpy(lin.astype(np.float32)), torch.from_numpy(rgba), H, band_rows, rank, world)
    if rank == 0:
        assert torch.equal(pf, frame) and torch.equal(p8, frame8)
    else:
        assert pf is None and p8 is None
    # the pipelined form of bench.py: start() of frame k + 1 completes the gather of frame k; frames differ per step
    from pathtrace_amd.dist import FilmGather
    fg = FilmGather(H, W, band_rows, rank, world, torch.device("cpu"))
    lin_t, rgba_t = torch.from_numpy(lin.astype(np.float32)), torch.from_numpy(rgba)
    got = []
    for k in range(3):
        fg.start(lin_t + float(k), (rgba_t + k).to(torch.uint8))
        if k:
            got.append(fg._last)                     # frame k - 1, completed by this start()
            fg._last = (None, None)
    got.append(fg.finish())
    assert fg.finish() == (None, None)               # nothing pending any more
    for k, (gl, g8) in enumerate(got):
        if rank == 0:
            assert torch.equal(gl, frame + float(k)) and torch.equal(g8, (frame8 + k).to(torch.uint8)), k
        else:
            assert gl is None and g8 is None
    # the pre-packed form (bench.py over RCCL since round 4): the renderer's film resolve writes the 16 B/pixel records into
    # fg.send itself (pt_render_device_packed) and start_prepacked() only launches the gather.  The send tile is double
    # buffered: frame k + 1 is written into the other tile WHILE the gather of frame k is in flight -- here the next tile is
    # filled before the previous frame is collected, and every frame must still come out whole.
    fg2 = FilmGather(H, W, band_rows, rank, world, torch.device("cpu"))
    n_rows = lin_t.shape[0]
    def fill(k):
        dst = fg2.send                                            # the tile the next frame goes into
        dst[:n_rows, :, :12] = (lin_t + float(10 * k)).contiguous().reshape(-1).view(torch.uint8).reshape(n_rows, W, 12)
        dst[:n_rows, :, 12:] = (rgba_t + 3 * k).to(torch.uint8)
        return dst.data_ptr()
    ptrs, got2 = [], []
    ptrs.append(fill(0)); fg2.start_prepacked()
    for k in range(1, 4):
        ptrs.append(fill(k))                                      # written while gather k - 1 is still pending
        fg2.start_prepacked()                                     # completes gather k - 1 first
        got2.append(fg2._last); fg2._last = (None, None)
    got2.append(fg2.finish())
    assert ptrs[0] != ptrs[1] and ptrs[0] == ptrs[2] and ptrs[1] == ptrs[3]      # two tiles, taken in turn
    for k, (gl, g8) in enumerate(got2):
        if rank == 0:
            assert torch.equal(gl, frame + float(10 * k)) and torch.equal(g8, (frame8 + 3 * k).to(torch.uint8)), k
        else:
            assert gl is None and g8 is None
    dist.barrier()
    if rank == 0:
        np.savez(out_path, lin=frame.numpy(), rgba=frame8.numpy())
    else:
        assert frame is None and frame8 is None
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,band_rows", [(2, 22, 4), (2, 16, 16), (2, 9, 1), (3, 23, 3), (4, 21, 2), (4, 5, 4)])
def test_ranks_gather_equals_single_render(pt, orc, tmp_path, world, H, band_rows):
    """world_size 2, 3 and 4 with ragged bands (the last band short, ranks that own fewer bands than others or
    nothing at all): the gathered frame equals the single-rank film bit for bit."""
    W = 20
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), H, W, band_rows, out), nprocs=world, join=True)
    got = np.load(out)
    cam = pt.camera_new(width=W, height=H)
    full, full8, _ = orc.render(cam, pt.builtin_scene(1), pt.default_params(spp=2), orc.F32, orc.ITERATIVE)
    assert np.array_equal(got["lin"], full.astype(np.float32))
    assert np.array_equal(got["rgba"], full8)


def test_gather_single_rank_reorders_nothing(pt):
    from pathtrace_amd.dist import gather_tiles
    t = torch.arange(5 * 3 * 2, dtype=torch.float32).reshape(5, 3, 2)
    assert torch.equal(gather_tiles(t, 5, 2, 0, 1), t)
