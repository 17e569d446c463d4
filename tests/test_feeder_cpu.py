"""The per-device host threads of the single-process multi-GPU form (pathtrace_amd/csrc/pt_feeder.h), on the CPU: jobs are
posted frame by frame to one worker per device the way pt_multi_render_device posts a frame; every job logs its start and
its end.  What pt_multi.cpp relies on:
  * a worker runs ITS jobs in the order they were posted and one at a time (a device's stream sees frame k before k + 1);
  * different workers run concurrently (that is the point: the per-device enqueue times overlap);
  * drain() returns only when every posted job has ended, and reports the first failure exactly once, while the jobs
    after a failed one still run (the devices' streams stay in step)."""
import ctypes as C

import numpy as np
import pytest

END = 1 << 63


def _run(pt, workers, frames, spin=2000, fail_at=-1):
    n = 2 * workers * frames
    log = (C.c_uint64 * max(n, 1))()
    cnt = C.c_uint32(0)
    rc = pt._lib.lib().pt_debug_feeder_selftest(workers, frames, spin, fail_at, log, C.byref(cnt))
    return rc, [int(v) for v in log[: cnt.value]]


@pytest.mark.parametrize("workers,frames", [(1, 5), (2, 40), (8, 25), (3, 0)])
def test_frames_stay_in_order_on_every_device_thread(pt, workers, frames):
    rc, log = _run(pt, workers, frames)
    assert rc == 0
    assert len(log) == 2 * workers * frames          # drain() waited for every job
    for w in range(workers):
        mine = [v for v in log if ((v & ~END) >> 32) == w]
        # start f, end f, start f + 1, end f + 1, ...: FIFO and never two jobs of one worker at once
        want = [x for f in range(frames) for x in ((w << 32) | f, END | (w << 32) | f)]
        assert mine == want


def test_device_threads_overlap(pt):
    """With 4 workers and long jobs some job starts while a job of ANOTHER worker is running (the log is not a sequence of
    start/end pairs) -- the enqueue costs of the devices overlap instead of adding up."""
    rc, log = _run(pt, 4, 30, spin=200000)
    assert rc == 0
    open_jobs, overlapped = set(), False
    for v in log:
        tag = v & ~END
        if v & END:
            open_jobs.discard(tag)
        else:
            overlapped = overlapped or len(open_jobs) > 0
            open_jobs.add(tag)
    assert overlapped and not open_jobs


def test_a_failed_enqueue_is_reported_once_and_later_frames_still_run(pt):
    workers, frames = 3, 6
    rc, log = _run(pt, workers, frames, fail_at=1 * frames + 2)      # worker 1, frame 2
    assert rc == 3                                                   # PT_ERR_HIP, what the job returned
    assert "failed as asked" in pt._lib.lib().pt_last_error().decode()
    assert len(log) == 2 * workers * frames                          # nothing was dropped
    rc, _ = _run(pt, workers, frames)                                # a fresh feeder starts clean
    assert rc == 0
