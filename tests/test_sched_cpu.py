"""The launch scheduler without a GPU.

pt_api.cpp's render_impl decides stream, buffer set, lane, exchange region and ordering events of every launch and resolve by
a PURE planner (csrc/pt_sched.h) and then executes the plan; pt_debug_sched_* runs the same planner on a scheduling state of
its own.  Here thousands of random job sequences -- lanes / queue form / split / profile / in_order / captured renders and
their replays / 1-4 batches / changing grids / failures half-way / synchronisations -- go through it and through the
happens-before simulator of tests/sched_sim.py, which asserts the invariants of DESIGN.md 3.  The two scheduling bugs of
round 4 (a GPU memory fault: 514c507^; 3 % of a film's samples silently lost: 41d6318^) are switched back on to show that
the simulator finds them without a GPU."""
import ctypes as C
import random

import pytest

import pathtrace_amd as pt
from pathtrace_amd._lib import PtSchedJob, PtSchedOp
from tests import sched_sim as S

CAP = 4096
FAULT_XCHG, FAULT_NO_WAIT = 1, 2


class Sched:
    def __init__(self):
        self.lib = pt._lib.lib()
        self.h = C.c_void_p()
        assert self.lib.pt_debug_sched_create(C.byref(self.h)) == 0
        self.buf = (PtSchedOp * CAP)()

    def close(self):
        self.lib.pt_debug_sched_destroy(self.h)

    def render(self, job, faults=0, fail_after=0xFFFFFFFF):
        n, lanes = C.c_uint32(), C.c_uint32()
        rc = self.lib.pt_debug_sched_render(self.h, C.byref(job), faults, fail_after, self.buf, CAP, C.byref(n), C.byref(lanes))
        assert rc in (0, 3), pt._lib.lib().pt_last_error()
        ops = []
        for i in range(n.value):
            o = PtSchedOp()
            C.memmove(C.byref(o), C.byref(self.buf[i]), C.sizeof(PtSchedOp))
            ops.append(o)
        return rc, ops, bool(lanes.value)

    def sync(self):
        assert self.lib.pt_debug_sched_sync(self.h, 1) == 0


def make_job(rng, form=None, **kw):
    """A job as render_impl derives it: `form` = what the scene and batch size select."""
    form = form or rng.choice(["regen", "regen", "regen", "split", "split", "queue_small", "queue_large", "regen_export"])
    j = PtSchedJob()
    j.n_batches = rng.choice([1, 1, 1, 2, 3, 4])
    j.regen_export = 1
    j.counter_words = 576
    j.cont_grid = 1024
    j.grid = rng.choice([2048, 8192, 24576])
    cap = rng.choice([1536, 1280])                       # 6 / 5 workgroups per CU x 256 CUs
    j.regen_capacity = cap
    # the whole device, or what a smaller image has chunks for (a rank's share): the grid changes from render to render
    j.regen_grid = rng.choice([cap, cap, cap // 2, cap // 8, 96, 7])
    j.fixed_grid = int(rng.random() < 0.1)
    j.profile = int(rng.random() < 0.12)
    j.in_order = int(rng.random() < 0.12)
    j.capturing = 0
    if form in ("regen", "split", "regen_export"):
        j.regen, j.hand_off = 1, 1
        j.split = int(form == "split")
        if form == "regen_export":
            j.regen_export = 16                          # PtTuning.export_below: the regenerating waves hand over
        if j.split:
            j.xchg_need = j.regen_grid * 4 * 640 + 1024
    elif form == "queue_large":
        j.hand_off = 1
    for k, v in kw.items():
        setattr(j, k, v)
    return j


def drive(seed, n_renders, faults=0, with_capture=True, with_failures=True):
    """One random sequence through the planner and the simulator; raises sched_sim.Violation."""
    rng = random.Random(seed)
    sch, sim = Sched(), S.Sim()
    graphs = []                                          # captured renders: (ops, n_batches, job)
    exact = True
    try:
        rid = 0
        for _ in range(n_renders):
            rid += 1
            r = rng.random()
            if with_capture and graphs and r < 0.08:     # replay of a captured graph on the caller's stream
                ops, nb, job = rng.choice(graphs)
                sim.run(("replay", rid), ops, nb, False, job)
            else:
                job = make_job(rng)
                if with_capture and r > 0.95:
                    job.capturing = 1
                fail_after = 0xFFFFFFFF
                if with_failures and rng.random() < 0.04:
                    fail_after = rng.randrange(0, 12)
                rc, ops, lanes = sch.render(job, faults, fail_after)
                if lanes and (job.profile or job.in_order or job.capturing):
                    raise S.Violation(f"L: render {rid} takes the lanes although profile / in_order / capturing is set")
                if job.capturing and rc == 0:
                    graphs.append((ops, job.n_batches, job))      # nothing runs during the capture
                    exact = False
                    sim.any_capture = True
                elif rc == 0:
                    sim.render(rid, ops, job.n_batches, lanes, job)
                else:                                    # failed half-way: what was enqueued ran, then the recovery's host sync
                    try:
                        sim.run(rid, ops, job.n_batches, lanes, job)
                    except S.Violation as e:
                        # a truncated render may leave a resolve without its launch etc.; only races count here
                        if not str(e).startswith(("S:", "P:")):
                            raise
                    sim.failed(rid)
            if rng.random() < 0.3:
                sim.sync(exact)
                sch.sync()
        sim.sync(exact)
        sch.sync()
        return sim.n_ops
    finally:
        sch.close()


def test_thousands_of_random_job_sequences_keep_the_invariants():
    total = 0
    for seed in range(3000):
        total += drive(seed, n_renders=random.Random(seed).randrange(2, 14))
    assert total > 100000        # the sequences were not trivially short


def test_long_pipelines_without_synchronisation():
    """40 renders back to back behind ONE synchronisation, forms mixed: the rotation of lanes and sets wraps many times."""
    for seed in range(200):
        drive(10_000 + seed, n_renders=40, with_failures=False)


def _first_violation(faults, seeds, **kw):
    for seed in seeds:
        try:
            drive(seed, n_renders=random.Random(seed).randrange(2, 14), faults=faults, **kw)
        except S.Violation as e:
            return seed, str(e)
    return None, None


def test_the_memory_fault_of_round_4_is_caught():
    """514c507^: the exchange region of a lane followed the grid of the CURRENT render, so regions of renders with different
    grids, in flight together, overlapped (a GPU memory fault on C1).  Deterministic minimal case + the random sequences."""
    rng = random.Random(1)
    for faults, expect in ((0, None), (FAULT_XCHG, "X:")):
        sch, sim = Sched(), S.Sim()
        err = None
        try:
            for rid, grid in enumerate([1280, 160, 1280, 96, 640, 1280]):          # whole device, a rank's share, ...
                job = make_job(rng, "split", n_batches=1, profile=0, in_order=0, fixed_grid=0, regen_capacity=1280, regen_grid=grid,
                               xchg_need=grid * 4 * 640 + 1024)
                rc, ops, lanes = sch.render(job, faults)
                assert rc == 0 and lanes
                sim.render(rid, ops, 1, lanes, job)
            sim.sync(True)
        except S.Violation as e:
            err = str(e)
        finally:
            sch.close()
        assert (err is None) if expect is None else (err is not None and err.startswith(expect)), err
    seed, msg = _first_violation(FAULT_XCHG, range(400), with_failures=False, with_capture=False)
    assert seed is not None and msg.startswith("X:"), (seed, msg)


def test_the_lost_samples_of_round_4_are_caught():
    """41d6318^: a lanes render enqueued behind a still-queued render WITHOUT lanes (queue form, profiled, in order) started
    on buffer set 0 beside it: its launch wrote the sample buffer the queued render had yet to resolve, and that render's
    resolve cleared the chunk counters under the running launch (3 % of a film's samples lost, silently)."""
    rng = random.Random(2)
    for faults, expect in ((0, None), (FAULT_NO_WAIT, "R:")):
        sch, sim = Sched(), S.Sim()
        err = None
        try:
            a = make_job(rng, "regen", n_batches=1, profile=0, in_order=0, fixed_grid=0)       # lanes: sets 0 ...
            b = make_job(rng, "regen", n_batches=1, profile=0, in_order=1, fixed_grid=0)       # in order: the caller's stream, set 0
            for rid, job in enumerate([a, a, a, b, a, a]):
                rc, ops, lanes = sch.render(job, faults)
                assert rc == 0 and lanes == (job is a)
                sim.render(rid, ops, 1, lanes, job)
                if rid == 2:
                    sim.sync(True); sch.sync()          # the rotation starts over: the next lanes render takes set 0 again
            sim.sync(True)
        except S.Violation as e:
            err = str(e)
        finally:
            sch.close()
        assert (err is None) if expect is None else (err is not None and err.startswith(expect)), err
    seed, msg = _first_violation(FAULT_NO_WAIT, range(400), with_failures=False, with_capture=False)
    assert seed is not None and msg[:2] in ("R:", "C:"), (seed, msg)


def test_a_failure_half_way_leaves_a_state_the_next_renders_are_safe_in():
    """Every possible failing operation of a three-batch lanes render and of a two-level queue-form render: the operations
    before it ran, the recovery waited for every stream; the renders after it keep every invariant (in particular C: the
    launch counters a cut-off render left dirty are cleared before they are used again)."""
    rng = random.Random(3)
    for form in ("regen", "split", "queue_large", "queue_small"):
        probe = Sched()
        job = make_job(rng, form, n_batches=3, profile=0, in_order=0, fixed_grid=0)
        probe.render(job)
        _, full, _ = probe.render(job)                   # the plan of the SECOND render: the one that will be cut short
        probe.close()
        assert len(full) >= 8
        for cut in range(len(full)):
            sch, sim = Sched(), S.Sim()
            try:
                rc, ops, lanes = sch.render(job); sim.render(0, ops, 3, lanes, job)            # a good render first
                rc, ops, lanes = sch.render(job, 0, cut)
                assert rc == 3 and len(ops) == cut + 1 and ops[-1].kind == S.K_HOST_SYNC
                try:
                    sim.run(1, ops, 3, lanes, job)
                except S.Violation as e:
                    assert str(e).startswith(("S:", "P:")), e
                sim.failed(1)
                for rid in (2, 3, 4):
                    rc, ops, lanes = sch.render(job); assert rc == 0
                    sim.render(rid, ops, 3, lanes, job)
                sim.sync(True)
            finally:
                sch.close()


def test_a_captured_render_clears_what_it_uses_itself():
    """nothing executes during a capture, so the host's "these words are zero" flags say nothing about the moment
    of a replay.  A graph captured while the counters were clean, replayed after a direct render failed half-way, must still
    start from zero counters: the captured render carries its own fills and leaves the flags alone."""
    rng = random.Random(4)
    for form in ("regen", "queue_large"):
        sch, sim = Sched(), S.Sim()
        try:
            job = make_job(rng, form, n_batches=1, profile=0, in_order=0, fixed_grid=0)
            rc, ops, lanes = sch.render(job); sim.render(0, ops, 1, lanes, job)
            sim.sync(True); sch.sync()                                        # counters and statistics are clean now
            cap = make_job(rng, form, n_batches=1, profile=0, in_order=0, fixed_grid=0, capturing=1)
            rc, graph, lanes = sch.render(cap)
            assert rc == 0 and not lanes
            assert any(o.kind == S.K_MEMSET_COUNTERS for o in graph) and not any(o.kind == S.K_MEMSET_STATS for o in graph)
            rc, ops, lanes = sch.render(job, 0, 5 if form == "regen" else 2)  # a direct render dies after its launch(es), before its resolve
            assert rc == 3 and any(o.kind == S.K_LAUNCH for o in ops) and not any(o.kind == S.K_RESOLVE for o in ops)
            try:
                sim.run(2, ops, 1, lanes, job)
            except S.Violation as e:
                assert str(e).startswith(("S:", "P:")), e
            sim.failed(2)
            sim.run(("replay", 3), graph, 1, False, cap)                      # the replay starts from whatever that left
            rc, ops, lanes = sch.render(job); sim.render(4, ops, 1, lanes, job)
            sim.sync(False)
        finally:
            sch.close()


def test_contexts_share_no_scheduling_state():
    """Two contexts driven in turn plan exactly what each plans alone (the planner has no globals)."""
    def plans(order):
        rngs = {k: random.Random(100 + k) for k in (0, 1)}
        sch = {k: Sched() for k in (0, 1)}
        out = {0: [], 1: []}
        for k in order:
            _, ops, lanes = sch[k].render(make_job(rngs[k]))
            out[k].append((lanes, [bytes(o) for o in ops]))
        for s in sch.values():
            s.close()
        return out
    alone = plans([0] * 30 + [1] * 30)
    mixed = plans([0, 1] * 30)
    assert alone == mixed


def test_the_planner_matches_the_documented_protocol_on_the_headline_case():
    """Three single-batch lanes renders back to back (the bench's timed region), then a fourth: launch k goes to lane k % 3
    and set k % 3, waits for set_free of k - 3 and lane_begun of k - 1; resolves sit on the caller's stream behind lane_done."""
    rng = random.Random(5)
    sch = Sched()
    try:
        job = make_job(rng, "regen", n_batches=1, profile=0, in_order=0, fixed_grid=0, regen_capacity=1536, regen_grid=1536)
        seen = []
        for k in range(7):
            rc, ops, lanes = sch.render(job)
            assert lanes
            launch = [o for o in ops if o.kind == S.K_LAUNCH]
            assert len(launch) == 1
            l = launch[0]
            assert (l.lane, l.set, l.stream) == (k % 3, k % 3, S.S_LANE0 + k % 3)
            assert l.core == 1536 * 11 // 24 and l.seq == k + 1 and not (l.flags & S.F_STATIC)
            waits = [(o.stream, o.event) for o in ops if o.kind == S.K_WAIT]
            assert (S.S_CALLER, 7 + k % 3) in waits                                   # resolve behind lane_done
            assert ((S.S_LANE0 + k % 3, 15 + k % 3) in waits) == (k >= 3)             # set_free of launch k - 3
            assert ((S.S_LANE0 + k % 3, 11 + (k - 1) % 3) in waits) == (k >= 1)       # lane_begun of launch k - 1
            res = [o for o in ops if o.kind == S.K_RESOLVE]
            assert len(res) == 1 and res[0].stream == S.S_CALLER and res[0].set == k % 3 and res[0].zero_words == 576
            seen.append(len(ops))
        # steady state: no fills in the stream (the resolve clears the counters, pt_sync the statistics)
        rc, ops, _ = sch.render(job)
        assert not any(o.kind in (S.K_MEMSET_STATS, S.K_MEMSET_COUNTERS) for o in ops)
    finally:
        sch.close()
