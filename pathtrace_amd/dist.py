"""Multi-GPU: image tiles shard over ranks, one gather of the framebuffer.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm,
"gloo" for the CPU tests).  Every pixel is an independent unit whose RNG key is a
pure function of (x, y) (src/main.rs:51) and whose result lands in its own film
slot (src/main.rs:58), so ranks render disjoint row bands with NO data-path
collective; the only exchange is one gather of the finished tiles to rank 0.
Bands are interleaved (rank g owns bands g, g+G, g+2G, ...) because the cost of a
pixel depends on what it sees.  The result is bitwise independent of G.
"""
import torch
import torch.distributed as dist

from .api import tile_row_indices


def _lib():
    from . import _lib as L
    return L.lib()


def _check(rc):
    from . import _lib as L
    L.check(rc)


def _stream_ptr(device):
    """torch's current stream on `device` as the hipStream_t the C ABI takes (0 = the default stream)."""
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def default_band_rows(height, world_size, target_bands_per_rank=8):
    """Band height giving every rank about `target_bands_per_rank` interleaved bands."""
    return max(1, height // max(1, world_size * target_bands_per_rank))


def gather_tiles(tile, height, band_rows, rank, world_size, group=None, dst=0):
    """Gather per-rank tiles [rows_r, W, C] to `dst` and place their rows in image order.

    One collective: dist.gather of equally padded tiles (row counts differ by at most
    one band).  Returns the full [height, W, C] frame on `dst`, None elsewhere."""
    rows = [tile_row_indices(height, band_rows, g, world_size) for g in range(world_size)]
    max_rows = max(len(r) for r in rows)
    assert tile.shape[0] == len(rows[rank]), (tile.shape, len(rows[rank]))
    if world_size == 1:
        frame = torch.empty((height,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        frame[torch.as_tensor(rows[0], device=tile.device)] = tile
        return frame
    pad = torch.zeros((max_rows,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    pad[: tile.shape[0]] = tile
    bufs = [torch.empty_like(pad) for _ in range(world_size)] if rank == dst else None
    dist.gather(pad, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    frame = torch.empty((height,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
    for g in range(world_size):
        if rows[g]:
            frame[torch.as_tensor(rows[g], device=tile.device)] = bufs[g][: len(rows[g])]
    return frame


class FilmGather:
    """The per-frame exchange with everything that does not change from frame to frame done once: row
    partition, padded send tile, receive buffer and the row permutation that puts the gathered band rows in
    image order.  Per frame on a GPU: the library's pack kernel (pt_film_pack), ONE dist.gather, the library's unpack
    kernel (pt_film_unpack) -- the kernels pt_multi_* runs around its ncclGather; on the CPU (gloo tests): two pack
    copies, the gather, one index_select, two unpack copies.  No Python loop over rows, no host-to-device copy.  start()/finish() split the exchange so that the gather of frame k
    overlaps the rendering of frame k + 1 (bench.py); calling the object does both at once.

    Both film planes travel together: the f32 linear plane [rows, W, 3] and the RGBA8 plane [rows, W, 4] are
    packed into a [rows, W, 16] byte tile (12 + 4 bytes per pixel).

    On a GPU the exchange has a stream of its own and a RING of send tiles (round 4).  The collective's kernel can be late:
    a rank that renders frame after frame keeps three regenerating launches in flight (pt_api.cpp: lanes), and RCCL's kernel
    -- one workgroup of 4 x 136 VGPRs -- is not placed beside them until the launches run out of work
    (profiles/r04/gather_kernel_beside_lanes.txt).  A wait for the gather in the render stream therefore stalled the whole
    pipeline once per frame.  Now the render stream only records `tile written`; gather, wait and row permutation are
    enqueued on the exchange stream; and the render stream waits for a gather only when it wants that gather's tile back,
    RING_MAX frames later.  finish() makes the caller's stream wait for the last frame."""

    RING_MAX = 8                # send tiles on a GPU (host tensors: 2) ...
    RING_BYTES = 1 << 30        # ... fewer when this would not hold that many

    def __init__(self, height, width, band_rows, rank, world_size, device, group=None, dst=0, always_collective=False):
        rows = [tile_row_indices(height, band_rows, g, world_size) for g in range(world_size)]
        self.height, self.width, self.rank, self.world, self.group, self.dst = height, width, rank, world_size, group, dst
        self.band_rows = band_rows
        self.my_rows = len(rows[rank])
        self.max_rows = max(len(r) for r in rows)
        # two send tiles: with pre-packed renders (start_prepacked) the render of frame k + 1 writes one while the gather of
        # frame k, running on the backend's stream, still reads the other
        self._streamed = torch.device(device).type == "cuda"
        n_send = 2
        if self._streamed:
            n_send = int(min(self.RING_MAX, max(2, self.RING_BYTES // max(1, self.max_rows * width * 16))))
        self._sends = [torch.zeros((self.max_rows, width, 16), dtype=torch.uint8, device=device) for _ in range(n_send)]
        self._cur, self._pending_buf = 0, None
        if self._streamed:
            self._xs = torch.cuda.Stream(device)             # gather, wait for it, row permutation
            self._sent = [None] * n_send                     # per send tile: the event after the gather that read it
            self._last_ev, self._frames, self._outs = None, 0, None
        self.recv = self.bufs = self.perm = None
        self._work, self._pending, self._last = None, False, (None, None)
        # world_size 1 needs no exchange; always_collective issues the gather anyway (bench.py --force-dist: the whole
        # N > 1 code path over the real backend on a one-GPU box)
        self._collective = world_size > 1 or always_collective
        if rank == dst:
            self.recv = torch.empty((world_size, self.max_rows, width, 16), dtype=torch.uint8, device=device)
            self.bufs = [self.recv[g] for g in range(world_size)]
            perm = [0] * height
            for g in range(world_size):
                for k, y in enumerate(rows[g]):
                    perm[y] = g * self.max_rows + k
            self.perm = torch.tensor(perm, dtype=torch.int64).to(device)
            if self._streamed:   # two frames of output, written in turn by the exchange stream
                self._outs = [(torch.empty((height, width, 3), dtype=torch.float32, device=device),
                               torch.empty((height, width, 4), dtype=torch.uint8, device=device)) for _ in range(2)]

    @property
    def send(self):
        """The send tile the next frame goes into ([max_rows, W, 16] bytes)."""
        return self._sends[self._cur]

    def start(self, lin, rgba):
        """Pack this rank's tile and launch the gather without waiting for it (async_op): the collective runs
        on the backend's own stream / thread while the caller renders the next frame.  A previous gather still in
        flight is completed first (its frame is then available from finish())."""
        if self._streamed:
            self._pack(lin, rgba)
            return self._start_streamed()
        if self._work is not None or self._pending:
            self._last = self.finish()
        self._pack(lin, rgba)
        self._pending_buf = self.send
        if self._collective:
            self._work = dist.gather(self.send, gather_list=self.bufs, dst=self.dst, group=self.group, async_op=True)
        self._pending = True

    def start_prepacked(self):
        """start() for a tile that is already in self.send: Context.render_packed_into(cam, prm, self.send.data_ptr()) made the
        film resolve write the 16 B/pixel records there (pt_render_device_packed), so no pack step runs.  The render must
        have been enqueued on torch's current stream of the device, where the gather is ordered.  Afterwards self.send is
        the OTHER tile: the caller may enqueue the next render into it at once -- this gather keeps reading the one it was
        given, and by the time that one is written again (two frames on) the start of the frame in between has waited
        for this gather.  (GPU tiles: a ring of RING_MAX tiles and an exchange stream, see the class comment.)"""
        if self._streamed:
            return self._start_streamed()
        if self._work is not None or self._pending:
            self._last = self.finish()
        self._pending_buf = self.send
        if self._collective:
            self._work = dist.gather(self._pending_buf, gather_list=self.bufs, dst=self.dst, group=self.group, async_op=True)
        self._cur ^= 1
        self._pending = True

    def _start_streamed(self):
        """GPU form of start(): the tile in self.send was written by work enqueued on torch's current stream.  Everything
        here is an enqueue; nothing waits on the host."""
        buf = self.send
        cur = torch.cuda.current_stream(buf.device)
        ready = torch.cuda.Event()
        ready.record(cur)
        with torch.cuda.stream(self._xs):
            self._xs.wait_event(ready)
            if self._collective:
                work = dist.gather(buf, gather_list=self.bufs, dst=self.dst, group=self.group, async_op=True)
                work.wait()                                  # the exchange stream waits for the backend's stream
            if self.rank == self.dst:
                src = self.recv.view(self.world * self.max_rows, self.width, 16) if self._collective else buf
                self._unpack(src, self._outs[self._frames % 2])
            sent = torch.cuda.Event()
            sent.record(self._xs)
        self._sent[self._cur] = self._last_ev = sent
        self._frames += 1
        self._cur = (self._cur + 1) % len(self._sends)
        if self._sent[self._cur] is not None:                # the tile the next frame goes into: its last gather has read it
            cur.wait_event(self._sent[self._cur])
        self._pending = True

    def finish(self):
        """Wait for the gather launched by start() and return (linear [H, W, 3] f32, rgba [H, W, 4] u8) on `dst`,
        (None, None) elsewhere.  Without a pending gather: the frame of the last one completed by start().
        GPU tiles: torch's current stream waits (the host does not); the tensors returned are the exchange's own two
        output frames, written in turn -- copy what must outlive the next two start() calls."""
        if self._streamed:
            if not self._pending:
                return None, None
            torch.cuda.current_stream(self._sends[0].device).wait_event(self._last_ev)
            self._pending = False
            return self._outs[(self._frames - 1) % 2] if self.rank == self.dst else (None, None)
        if not self._pending:
            last, self._last = self._last, (None, None)
            return last
        if self._work is not None:
            self._work.wait()          # nccl: the current stream waits for the collective; gloo: the host does
            self._work = None
        self._pending = False
        if self.rank != self.dst:
            return None, None
        return self._unpack(self.recv.view(self.world * self.max_rows, self.width, 16) if self._collective else self._pending_buf)

    def _pack(self, lin, rgba):
        """This rank's tile -> self.send (16 bytes per pixel)."""
        n, w = self.my_rows, self.width
        assert lin.shape[0] == n and rgba.shape[0] == n, (lin.shape, rgba.shape, n)
        if self.send.is_cuda:
            # the library's pack kernel (pt_film_pack: 16-byte stores) on torch's current stream, where the gather is ordered
            lin, rgba = lin.contiguous(), rgba.contiguous()
            assert lin.dtype == torch.float32 and rgba.dtype == torch.uint8
            with torch.cuda.device(self.send.device):
                _check(_lib().pt_film_pack(_stream_ptr(self.send.device), lin.data_ptr(), rgba.data_ptr(), n * w, self.send.data_ptr()))
        else:
            self.send[:n, :, :12] = lin.contiguous().reshape(-1).view(torch.uint8).reshape(n, w, 12)
            self.send[:n, :, 12:] = rgba

    def _unpack(self, src, out=None):
        """Gathered padded tiles [world * max_rows, W, 16] -> (linear [H, W, 3] f32, rgba [H, W, 4] u8) in image order."""
        w = self.width
        if src.is_cuda:
            # the library's unpack kernel (pt_film_unpack): gathered padded tiles -> both film planes in image order
            if out is not None:
                lin_full, rgba_full = out
            else:
                lin_full = torch.empty((self.height, w, 3), dtype=torch.float32, device=src.device)
                rgba_full = torch.empty((self.height, w, 4), dtype=torch.uint8, device=src.device)
            with torch.cuda.device(src.device):
                _check(_lib().pt_film_unpack(_stream_ptr(src.device), src.data_ptr(), w, self.height, self.band_rows, self.world,
                                             self.max_rows, lin_full.data_ptr(), rgba_full.data_ptr()))
            return lin_full, rgba_full
        frame = src.index_select(0, self.perm)
        lin_full = frame[..., :12].contiguous().reshape(-1).view(torch.float32).reshape(self.height, w, 3)
        return lin_full, frame[..., 12:].contiguous()

    def __call__(self, lin, rgba):
        """start() + finish(): the blocking form."""
        self.start(lin, rgba)
        return self.finish()


_plans = {}


def gather_film(lin, rgba, height, band_rows, rank, world_size, group=None, dst=0):
    """ONE collective for both film planes (FilmGather; the plan is cached per geometry and device).
    Returns (linear [H, W, 3] f32, rgba [H, W, 4] u8) on `dst`, (None, None) elsewhere."""
    key = (height, lin.shape[1], band_rows, rank, world_size, str(lin.device), id(group), dst)
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = FilmGather(height, lin.shape[1], band_rows, rank, world_size, lin.device, group, dst)
    out = plan(lin, rgba)
    if plan._streamed and out[0] is not None:        # the plan's own output frames are written again two frames on
        out = (out[0].clone(), out[1].clone())
    return out


def render_distributed(ctx, cam, params, rank, world_size, band_rows=None, group=None, want_rgba=True):
    """Render this rank's interleaved bands on its GPU and gather the frame(s) on rank 0."""
    params.band_rows = band_rows or default_band_rows(cam.height, world_size)
    params.band_index = rank
    params.band_count = world_size
    lin, rgba = ctx.render(cam, params, want_rgba=want_rgba)
    if want_rgba:
        return gather_film(lin, rgba, cam.height, params.band_rows, rank, world_size, group)
    return gather_tiles(lin, cam.height, params.band_rows, rank, world_size, group), None
