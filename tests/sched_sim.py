"""A happens-before simulator for the stream operations the launch planner emits (csrc/pt_sched.h, exported through
pt_debug_sched_*).  Test infrastructure; it shares NO code with the planner -- it only knows what an operation MEANS:

  * streams are FIFOs; `record(e)` on a stream snapshots "everything enqueued on that stream so far, and what that waited
    for"; `wait(e)` makes what follows on the waiting stream happen after the snapshot `e` held AT THE TIME OF THE CALL
    (hipStreamWaitEvent semantics; a wait for an event never recorded is a no-op);
  * a launch writes the sample buffer of its set, takes its chunks from (or hands its tails over through) the set's launch
    counters, works in its exchange region, adds to the statistics words; a resolve reads the sample buffer, clears the
    set's counters, adds the batch to the film sums and, for the last batch, writes the outputs; fills clear what they name;
  * pt_sync waits for the CALLER's stream only.

What it asserts are the invariants of DESIGN.md section 3 ("Lanes: protocol"):
  R  no two operations that touch the same buffer set / counters / statistics / exchange region / queue / film, one of
     them writing, are unordered (every such pair is ordered by happens-before);
  C  a launch that uses a set's launch counters finds them zero (cleared by a fill or by the resolve that used them last,
     with nothing in between);
  S  a resolve reads exactly the samples of its own (render, batch); the film sums receive a render's batches in order;
  D  when the caller's stream is complete, everything the context enqueued is ("complete when the context's stream is");
  X  exchange regions of launches that may run concurrently are disjoint (it is R on intervals);
  P  the spare-workgroup word is published before the launch that reads it, numbers only grow; a profiling event pair
     brackets exactly one launch on its stream and is not reused before the statistics are collected;
  L  lanes are off for profiled, in-order and captured renders; a lanes launch deals no chunks statically;
  T  the statistics a collection reads are exactly those of the renders enqueued since the last collection;
  E  nothing waits for an event that was never recorded.
"""
import collections

K_HOST_SYNC, K_MEMSET_STATS, K_MEMSET_COUNTERS, K_RECORD, K_WAIT, K_POST, K_LAUNCH, K_RESOLVE = range(8)
F_REGEN, F_SPLIT, F_HANDOFF, F_STATIC, F_PRIMARY, F_LANES = 1, 2, 4, 8, 16, 32
S_CALLER, S_SIDE, S_LANE0 = 0, 1, 2
EV_POOL = 19
N_STREAMS = 2 + 4


class Violation(AssertionError):
    pass


class Sim:
    def __init__(self):
        self.pos = [0] * N_STREAMS                    # operations enqueued per stream
        self.tail = [[0] * N_STREAMS for _ in range(N_STREAMS)]   # vector clock of each stream's last operation
        self.floor = [0] * N_STREAMS                  # everything up to here is complete (host synchronisation)
        self.events = {}                              # event key -> vector clock snapshot
        # resource -> {"w": (stream, pos, what) of the last writer, "r": [readers / accumulators since]}
        self.res = collections.defaultdict(lambda: {"w": None, "r": []})
        self.xchg = []                                # (off, len, stream, pos, what) of split launches since the last host sync
        self.counters_clean = collections.defaultdict(bool)       # set -> known zero (a fresh buffer holds anything)
        self.lsamp_tag = {}                           # set -> (render, batch) whose samples it holds
        self.film = None                              # (render, batches added)
        self.stats_launches = []                      # renders whose launches added to the statistics since they were cleared
        self.stats_dirty_unknown = True               # a fresh buffer holds anything
        self.pending_renders = []                     # renders enqueued since the statistics were last collected
        self.posted = 0
        self.pool_used = set()
        self.n_ops = 0
        self.any_capture = False

    # ---- happens-before
    def _enqueue(self, stream, extra=None):
        self.pos[stream] += 1
        clk = [max(a, b) for a, b in zip(self.tail[stream], self.floor)]
        if extra is not None:
            clk = [max(a, b) for a, b in zip(clk, extra)]
        clk[stream] = self.pos[stream]
        self.tail[stream] = clk
        self.n_ops += 1
        return clk

    @staticmethod
    def _hb(a, clk):                                  # a = (stream, pos, what) happens before the operation with clock clk
        return clk[a[0]] >= a[1]

    def _access(self, key, mode, stream, clk, what):
        """mode 'w' exclusive, 'r' read, 'a' commuting accumulation"""
        r = self.res[key]
        me = (stream, clk[stream], what)
        if r["w"] is not None and not self._hb(r["w"], clk):
            raise Violation(f"R: {what} touches {key} unordered against {r['w'][2]}")
        if mode == "w":
            for o in r["r"]:
                if not self._hb(o, clk):
                    raise Violation(f"R: {what} writes {key} unordered against {o[2]}")
            r["w"], r["r"] = me, []
        else:
            if mode == "r":
                for o in r["r"]:
                    if o[3] == "a" and not self._hb(o, clk):
                        raise Violation(f"R: {what} reads {key} unordered against {o[2]}")
            r["r"].append(me + (mode,))

    def host_sync_all(self):
        self.floor = [max(self.floor[s], self.pos[s]) for s in range(N_STREAMS)]
        self.xchg = []

    # ---- one render's operations, enqueued now (a replay of a captured render enqueues them again)
    def run(self, render_id, ops, n_batches, lanes, job):
        profile_open = {}
        batch_levels = collections.Counter((o.batch) for o in ops if o.kind == K_LAUNCH)
        last_pool_begin = None
        for i, o in enumerate(ops):
            what = f"render {render_id} op {i} kind {o.kind} stream {o.stream} set {o.set} batch {o.batch}"
            if o.kind == K_HOST_SYNC:
                self.host_sync_all()
                continue
            if o.kind == K_POST:
                if o.seq <= self.posted:
                    raise Violation(f"P: {what} publishes {o.seq} after {self.posted}")
                self.posted = o.seq
                continue
            if o.kind == K_WAIT:
                key = (o.event, o.pool if o.event == EV_POOL else 0)
                if key not in self.events:
                    raise Violation(f"E: {what} waits for event {key}, which was never recorded")
                self._enqueue(o.stream, self.events[key])
                continue
            clk = self._enqueue(o.stream)
            if o.kind == K_RECORD:
                key = (o.event, o.pool if o.event == EV_POOL else 0)
                self.events[key] = list(clk)
                if o.event == EV_POOL:
                    if o.pool in self.pool_used:
                        raise Violation(f"P: {what} reuses profiling event {o.pool} before the statistics were collected")
                    self.pool_used.add(o.pool)
                    if o.pool % 2 == 0:
                        last_pool_begin = (o.pool, o.stream, i)
                    else:
                        if last_pool_begin is None or last_pool_begin[0] + 1 != o.pool or last_pool_begin[1] != o.stream or \
                                last_pool_begin[2] + 2 != i or ops[i - 1].kind != K_LAUNCH:
                            raise Violation(f"P: {what}: a profiling event pair must bracket exactly one launch on its stream")
                        last_pool_begin = None
            elif o.kind == K_MEMSET_STATS:
                self._access(("stats",), "w", o.stream, clk, what)
                self.stats_launches = []
                self.stats_dirty_unknown = False
            elif o.kind == K_MEMSET_COUNTERS:
                self._access(("counters", o.set), "w", o.stream, clk, what)
                self.counters_clean[o.set] = True
            elif o.kind == K_LAUNCH:
                regen = bool(o.flags & F_REGEN)
                if bool(o.flags & F_LANES) != (o.stream >= S_LANE0):
                    raise Violation(f"L: {what}: lanes flag and stream disagree")
                if (o.flags & F_LANES) and (o.flags & F_STATIC):
                    raise Violation(f"L: {what}: a lanes launch deals chunks statically")
                if (o.flags & F_LANES) and not lanes:
                    raise Violation(f"L: {what}: lane stream used by a render that does not take the lanes")
                if o.core:
                    if o.seq != self.posted:
                        raise Violation(f"P: {what}: launch {o.seq} starts with {self.posted} published")
                self._access(("lsamp", o.set), "w", o.stream, clk, what)
                self.lsamp_tag[o.set] = (render_id, o.batch)
                self._access(("stats",), "a", o.stream, clk, what)
                self.stats_launches.append(render_id)
                two_levels = batch_levels[o.batch] == 2
                if o.flags & F_HANDOFF:
                    self._access(("counters", o.set), "w", o.stream, clk, what)
                    if o.level == 0:
                        if not self.counters_clean[o.set]:
                            raise Violation(f"C: {what} starts with launch counters of set {o.set} that are not known to be zero")
                        self.counters_clean[o.set] = False
                    if two_levels:
                        self._access(("ovf", o.ovf_par), "w" if o.level == 0 else "r", o.stream, clk, what)
                if o.flags & F_SPLIT:
                    if o.xchg_len == 0:
                        raise Violation(f"X: {what}: split launch without an exchange region")
                    for (off, ln, st, ps, w2) in self.xchg:
                        if off < o.xchg_off + o.xchg_len and o.xchg_off < off + ln and not self._hb((st, ps, w2), clk):
                            raise Violation(f"X: {what} region [{o.xchg_off}, +{o.xchg_len}) overlaps [{off}, +{ln}) of {w2}, unordered")
                    self.xchg.append((o.xchg_off, o.xchg_len, o.stream, clk[o.stream], what))
                if not regen or o.level > 0:
                    self._access(("queue", o.own_queue), "w", o.stream, clk, what)
            elif o.kind == K_RESOLVE:
                self._access(("lsamp", o.set), "r", o.stream, clk, what)
                if self.lsamp_tag.get(o.set) != (render_id, o.batch):
                    raise Violation(f"S: {what} reads samples of {self.lsamp_tag.get(o.set)}")
                if o.zero_words:
                    self._access(("counters", o.set), "w", o.stream, clk, what)
                    self.counters_clean[o.set] = True
                if n_batches > 1:
                    self._access(("film",), "w", o.stream, clk, what)
                    want = (render_id, o.batch)
                    have = (render_id, 0) if o.batch == 0 else self.film
                    if have != want:
                        raise Violation(f"S: {what}: film sums hold {self.film}")
                    self.film = (render_id, o.batch + 1)
                if o.batch + 1 == n_batches:
                    self._access(("out", render_id), "w", o.stream, clk, what)
            else:
                raise Violation(f"unknown operation kind {o.kind}")
        if lanes and (job.profile or job.in_order or job.capturing):
            raise Violation(f"L: render {render_id} takes the lanes although profile / in_order / capturing is set")

    def render(self, render_id, ops, n_batches, lanes, job):
        self.run(render_id, ops, n_batches, lanes, job)
        self.pending_renders.append(render_id)

    def failed(self, render_id):
        """the render failed half-way and the recovery ran (its ops ended with a host sync): nothing is known to be clean"""
        self.counters_clean.clear()
        self.stats_dirty_unknown = True
        self.pending_renders = []
        self.pool_used = set()

    # ---- pt_sync: waits for the caller's stream; collects the statistics if renders are pending
    def sync(self, exact_stats):
        clk = [max(a, b) for a, b in zip(self.tail[S_CALLER], self.floor)]
        for s in range(N_STREAMS):
            if clk[s] < self.pos[s]:
                raise Violation(f"D: the caller's stream is complete but stream {s} has {self.pos[s] - clk[s]} operation(s) it never waited for")
        self.host_sync_all()
        if self.pending_renders:
            if exact_stats:
                if self.stats_dirty_unknown:
                    raise Violation("T: statistics collected from words that were never cleared")
                if sorted(set(self.stats_launches)) != sorted(set(self.pending_renders)):
                    raise Violation(f"T: the collection reads launches of renders {sorted(set(self.stats_launches))}, "
                                    f"enqueued since the last one: {sorted(set(self.pending_renders))}")
            # pt_sync clears the words for the next renders: a fill on the caller's stream, behind nothing but that stream
            clk = self._enqueue(S_CALLER)
            self._access(("stats",), "w", S_CALLER, clk, "pt_sync's clearing of the statistics")
            self.stats_launches = []
            self.stats_dirty_unknown = False
            self.pending_renders = []
            self.pool_used = set()
