// pt_sched.h -- the launch scheduler of a context as a PURE function: (scheduling state, job) -> list of stream operations.
//
// render() == everything src/main.rs:43-60 does for a tile; on the device that is, per sample batch, one or two path-kernel
// launches and a film resolve, spread over the context's streams ("Lanes").  Which stream an operation goes to,
// which buffer set / exchange region / counters it touches and which events order it are decided HERE, with no HIP call and no
// context in sight; pt_api.cpp's render_impl prepares the job (sizes, allocations, the launch arguments), calls plan() on a COPY
// of the context's State, issues the operations one by one (execute) and commits the copy only when the last one was enqueued.
//
// Because it is pure it can be checked without a GPU: pt_debug_sched_* (include/pathtrace_amd.h) exposes it, and
// tests/test_sched_cpu.py drives thousands of random job sequences through it and a happens-before simulator that knows
// nothing of this file but the meaning of the operations -- the invariants of DESIGN.md 3 are what it asserts.  The two
// scheduling bugs of round 4 can be switched back on (Faults) to show that the simulator finds them.
//
// No HIP, no allocation besides the vector of operations.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#ifndef PT_LANE_OVERLAP
#define PT_LANE_OVERLAP 1
#endif
#ifndef PT_LANES
#define PT_LANES 3
#endif
#ifndef PT_SETS
#define PT_SETS 3
#endif
#ifndef PT_LANE_GRID24
#define PT_LANE_GRID24 11
#endif
#ifndef PT_LANE_GRID24_SPLIT
#define PT_LANE_GRID24_SPLIT 12
#endif

namespace ptsched {

constexpr bool kLaneOverlap = PT_LANE_OVERLAP != 0;
// streams the regenerating launches of consecutive batches take in turn (2 or 3)
constexpr int kLanes = PT_LANES;
// Buffer sets (sample buffer + launch counters) the lanes' launches rotate through: launch k waits for the resolve of launch
// k - kSets.  With three, launch k + 3 waits for resolve k, which gets few wave slots beside the resident launches (146 us of work
// take ~0.9 ms).  A fourth set, so that it need not, was measured and LOSES: the resolve then competes with one more pending launch
// (C2 5.67 -> 6.1 ms per step, C1 7.15 -> 7.3; profiles/r04/ab_four_buffer_sets.txt).
constexpr int kSets = PT_SETS;
static_assert(kLanes >= 2 && kLanes <= 4, "PT_LANES: 2 .. 4");
static_assert(kSets >= 3 && kSets <= 4, "PT_SETS: 3 or 4");
// core of a launch that overlaps its neighbours, in 24ths of what the device holds: k_paths_regen / k_paths_regen_split (5 workgroups
// per CU: C1 7.26 ms per step at 11, 7.13 at 12, 7.18 at 13; profiles/r04/ab_lane_core_size.txt)
constexpr uint32_t kLaneGrid24 = PT_LANE_GRID24, kLaneGrid24Split = PT_LANE_GRID24_SPLIT;
constexpr uint32_t kMaxProfiledLaunches = 1u << 14;      // bound of the HIP-event pool (profile = 1) between two collections of the statistics

// ---- streams and events of a context (indices; pt_api.cpp maps them to its hipStream_t / hipEvent_t)
enum : uint32_t { kStreamCaller = 0, kStreamSide = 1, kStreamLane0 = 2 };            // kStreamLane0 + lane
enum : uint32_t {
    kEvBegin = 0, kEvEnd = 1,       // first / last operation of the renders since the statistics were last collected (total_ms)
    kEvPre = 2,                     // what a lanes render put on the caller's stream before its first launch
    kEvL0 = 3,                      // + parity: level-0 launch of an overlapped (queue form, several batches) batch is through
    kEvResolved = 5,                // + parity: ... its tail (continuation launch, resolve) on the side stream is through
    kEvLaneDone = 7,                // + lane: the lane's last launch is through
    kEvLaneBegun = 11,              // + lane: the lane's last launch has been handed to the device
    kEvSetFree = 15,                // + set: the resolve that read the buffer set last is through
    kEvPool = 19,                   // Op.pool = index into the profiling event pool (launch begin / end pairs)
    kEvCount = 20
};

enum : uint32_t {
    kOpHostSync = 0,      // the host waits for every stream of the context (only before the exchange memory grows)
    kOpMemsetStats = 1,   // stream: clear the statistics words
    kOpMemsetCounters = 2,// stream, set: clear the launch counters of a buffer set
    kOpRecord = 3,        // stream, event (+ pool)
    kOpWait = 4,          // stream, event
    kOpPost = 5,          // host: publish seq as "launches enqueued so far" (the word spare workgroups read)
    kOpLaunch = 6,        // stream: a path-kernel launch
    kOpResolve = 7,       // stream: the film resolve of a batch
};
// Op.flags of a launch
enum : uint32_t {
    kLaunchRegen = 1u,        // regenerating level-0 kernel: takes its chunks from the set's counters, paths live in registers
    kLaunchSplit = 2u,        // ... k_paths_regen_split: uses the exchange region [xchg_off, xchg_off + xchg_len)
    kLaunchHandOff = 4u,      // hands tails over / takes them up through the set's counter word and the hand-over queue ovf_par
    kLaunchStaticDeal = 8u,   // a regenerating launch that runs in order deals a quarter of its chunks statically
    kLaunchPrimary = 16u,     // level-0 launch (statistics: primary_*)
    kLaunchLanes = 32u,       // goes to a lane stream
};

// One stream operation.  Plain words so that the debug entry can hand the list to a test as it is.
struct Op {
    uint32_t kind, stream, event, pool;
    uint32_t set;          // launch / resolve / counter memset: buffer set (sample buffer + launch counters)
    uint32_t lane;         // launch: lane (lanes renders), else 0
    uint32_t level;        // launch: 0 = level-0, 1 = continuation launch
    uint32_t own_queue;    // launch: continuation launch of an overlapped batch (its own queue and scratch)
    uint32_t ovf_par;      // launch: which hand-over queue
    uint32_t batch;        // launch / resolve: sample batch of the render
    uint32_t grid;         // launch: workgroups (four-wave units)
    uint32_t seq, core;    // launch with spare workgroups: its number and its core size; post: the number published
    uint32_t flags;        // launch: kLaunch*
    uint32_t zero_words;   // resolve: launch-counter words of `set` its workgroup 0 clears
    uint32_t reserved;
    uint64_t xchg_off, xchg_len;   // launch (split): exchange region in float4
};

// Scheduling state of a context between renders.
struct State {
    uint32_t set_next = 0, lane_next = 0;     // rotation of buffer sets / lanes
    uint32_t lane_seq = 0;                    // number of the last launch with spare workgroups
    uint8_t lane_used[4] = {}, set_used[4] = {};   // the lane's / set's events have been recorded at least once
    // the device-side words are known to be zero: the statistics (cleared when they were last collected), the launch counters
    // of a set (cleared by the resolve of the batch that used them last) -- a render then needs no memset in the stream
    uint8_t counters_clean[4] = {};
    uint8_t stats_clean = 0;
    uint8_t stats_pending = 0;                // renders enqueued since the statistics were last collected
    uint8_t stream_work_since_lanes = 0;      // a render without lanes has been enqueued since the last one with
    uint8_t captured_any = 0;                 // a render has been captured into a graph: replays may run at any time, so the
                                              // statistics words are never known to be clean again
    uint32_t profiled = 0;                    // event slots (launch begin / end pairs) in use since the statistics were last collected
    uint64_t xchg_stride = 0;                 // float4 per lane of exchange memory (grows only while every stream is idle)
};

// What render_impl has derived from the scene, the parameters and the device before anything is enqueued.
struct Job {
    uint32_t n_batches = 1;
    uint32_t regen = 0;          // level-0 launches take a regenerating kernel
    uint32_t split = 0;          // ... k_paths_regen_split
    uint32_t hand_off = 0;       // large batch: tails are handed over (queue form) / chunk counters are used (regenerating forms)
    uint32_t regen_export = 1;   // regenerating waves hand over below this many live paths (<= 1: they run dry themselves)
    uint32_t profile = 0;        // PtRenderParams.profile
    uint32_t in_order = 0;       // PtTuning.in_order, or the render could not get its buffer sets (fall back, do not fail)
    uint32_t capturing = 0;      // the caller's stream is being captured into a graph
    uint32_t grid = 1;           // queue-form level-0 launch
    uint32_t regen_grid = 1;     // regenerating launch
    uint32_t cont_grid = 1;      // continuation launch
    uint32_t regen_capacity = 0; // workgroups of the regenerating kernel the device holds at once
    uint32_t fixed_grid = 0;     // the caller fixed the regenerating grid (params.workgroups / tuning.regen_workgroups): no spare workgroups
    uint32_t counter_words = 0;  // launch-counter words of a set a regenerating launch uses (kCountStride)
    uint64_t xchg_need = 0;      // float4 of exchange memory one launch of this render needs (split)
};

// The scheduling bugs of round 4, for tests/test_sched_cpu.py to show that it catches them (never set by the product).
enum : uint32_t {
    kFaultXchgStridePerRender = 1u,      // 514c507^: a lane's exchange region at lane * (this render's need) -- regions of renders with different grids overlap
    kFaultNoWaitAfterStreamWork = 2u,    // 41d6318^: a lanes render does not wait for a queue-form render still queued on the caller's stream
    kFaultResolveBeforeLaunch = 4u,      // a resolve does not wait for its lane's launch (a deliberately broken build: -DPT_SCHED_FAULTS=4
                                         // must make bench.py exit non-zero, profiles/r05/broken_build_is_caught.txt)
};

struct Plan {
    std::vector<Op> ops;
    uint32_t launches = 0, primary_launches = 0;
    bool lanes = false, overlap = false, profile = false, accumulate = false;
};

inline bool takes_lanes(const Job& j) {
    return kLaneOverlap && j.regen && j.regen_export <= 1u && !j.profile && !j.in_order && !j.capturing;
}
// buffer sets a render will use, in order (prepare: allocate them before anything is enqueued)
inline void sets_of(const State& s, const Job& j, uint32_t out[4], uint32_t* n) {
    *n = 0;
    if (takes_lanes(j)) {
        for (uint32_t k = 0; k < std::min<uint32_t>(j.n_batches, (uint32_t)kSets); ++k) out[(*n)++] = (s.set_next + k) % (uint32_t)kSets;
    } else {
        out[(*n)++] = 0;
        if (j.n_batches > 1 && !j.regen) out[(*n)++] = 1;
    }
}

// Plans one render.  `s` is advanced to the state after the render (the caller commits it once the operations are enqueued).
inline Plan plan(State& s, const Job& j, uint32_t faults = 0) {
    Plan p;
    auto push = [&p](uint32_t kind, uint32_t stream, uint32_t event = 0, uint32_t pool = 0) -> Op& {
        Op o{};
        o.kind = kind; o.stream = stream; o.event = event; o.pool = pool;
        p.ops.push_back(o);
        return p.ops.back();
    };
    const uint32_t n_batches = std::max(1u, j.n_batches);
    // Regenerating launches whose waves run dry themselves take the lanes in turn: the launch of the next batch -- the next render's,
    // when renders are enqueued back to back -- goes to the next lane's stream and starts as soon as that stream and its buffer set
    // are free, i.e. while earlier launches are still running.  Resolves stay on the caller's stream, in order.  Not while that stream
    // is being captured into a graph (the capture takes the in-order form), nor with profile = 1 (launches that overlap cannot be
    // timed one by one), nor with PtTuning.in_order.
    const bool lanes = takes_lanes(j);
    // Multi-batch renders in the queue form overlap the tail of batch k (continuation launch, resolve: side stream) with the body
    // of batch k + 1.  Not beside a regenerating level-0 launch (its waves hold every wave slot of the device).
    const bool overlap = n_batches > 1 && !j.regen;
    const bool two_sets = overlap || lanes;
    const bool accumulate = s.stats_pending != 0;
    if (!accumulate) s.profiled = 0;
    // (a caller that pipelines profiled renders without ever collecting the statistics stops adding event pairs at kMaxProfiledLaunches;
    // a captured render carries none: its events would be re-recorded by every replay)
    const bool profile = j.profile != 0 && !j.capturing && s.profiled + 2u * n_batches <= kMaxProfiledLaunches;
    const uint32_t launch_words = j.hand_off ? (j.regen ? j.counter_words : 1u) : 0u;
    const uint32_t n_levels = j.hand_off && !(j.regen && j.regen_export <= 1u) ? 2u : 1u;
    // With lanes the first kLaneGrid24 / 24 of a launch's workgroups -- LESS than half of what the device holds -- are its core, the
    // others spare: they end as soon as they get a slot and see two launches enqueued behind theirs.
    uint32_t regen_core = 0;
    if (lanes && !j.fixed_grid) regen_core = std::max(1u, j.regen_capacity * (j.split ? kLaneGrid24Split : kLaneGrid24) / 24u);
    p.lanes = lanes; p.overlap = overlap; p.profile = profile; p.accumulate = accumulate;

    // One exchange region per lane (a launch may be in flight on each).  The lane stride is a property of the CONTEXT, not of this
    // render: launches of earlier renders may still be using their regions, and the memory a launch needs differs from render to
    // render.  It only ever grows, and only with every stream idle.
    uint64_t stride = s.xchg_stride;
    if (j.regen && j.split) {
        if (faults & kFaultXchgStridePerRender) {
            stride = j.xchg_need;                       // (the buffer is still sized for the largest stride seen)
            s.xchg_stride = std::max(s.xchg_stride, stride);
        } else if (j.xchg_need > s.xchg_stride) {
            push(kOpHostSync, kStreamCaller);
            s.xchg_stride = stride = j.xchg_need;
        }
    }

    // The device-side words are normally zero already.  Only a fresh buffer, a new scene or a render that failed half-way leaves
    // something to clear here -- and a CAPTURED render always clears the launch counters it uses itself (nothing executes during the
    // capture, so the flags say nothing about the moment of a replay) and leaves the flags alone.
    bool lanes_wait_pre = false;       // the lanes' launches of this call start behind what this call puts on the caller's stream first
    // The STATISTICS words a captured render leaves alone: its replays add to whatever period they run in (pt_sync accepts multiples
    // of a captured render's samples), and a fill inside the graph would wipe the counts of the direct renders of that period.
    // They are zero whenever no render is pending -- cleared at their collection, at a scene upload and after a failure -- except
    // that replays may have run since: once a render has been captured, a direct render that starts a period clears them itself.
    if (!accumulate) {
        if (!j.capturing && (s.captured_any || !s.stats_clean)) push(kOpMemsetStats, kStreamCaller);
        if (!j.capturing) s.stats_clean = 0;
        push(kOpRecord, kStreamCaller, kEvBegin);
        lanes_wait_pre = true;         // (the statistics' clearing -- here or at their collection -- is on that stream)
    }
    for (uint32_t par = 0; par < (lanes ? (uint32_t)kSets : two_sets ? 2u : 1u); ++par)
        if (launch_words && (j.capturing || !s.counters_clean[par])) {
            push(kOpMemsetCounters, kStreamCaller).set = par;
            if (!j.capturing) s.counters_clean[par] = 1;
            lanes_wait_pre = true;
        }
    // A render that does NOT take the lanes runs on the caller's stream and uses buffer sets 0 / 1 and their counters there.  Lane
    // launches are ordered behind RESOLVES of lane launches only, so the first lanes render after such a render waits for the
    // caller's stream as it stands now -- otherwise its launch could write the sample buffer a queued queue-form render has yet to
    // resolve, or have its chunk counters cleared by that render's resolve (round 4: 3 % of a film's samples lost).
    // (Once a render of this context has been captured into a graph, a replay -- which the library does not see -- may be such a
    // render: every lanes render then waits for the caller's stream.  Found by tests/test_sched_cpu.py.)
    if (lanes && (s.stream_work_since_lanes || s.captured_any) && !(faults & kFaultNoWaitAfterStreamWork)) lanes_wait_pre = true;
    s.stream_work_since_lanes = lanes ? 0 : 1;
    if (lanes && lanes_wait_pre) push(kOpRecord, kStreamCaller, kEvPre);
    bool lane_waited_pre[4] = {};
    const uint32_t ev0 = s.profiled;      // first free event slot
    const uint32_t side = overlap ? kStreamSide : kStreamCaller;
    if (overlap) {      // the side stream starts after whatever the caller's stream holds so far (previous renders, film state)
        push(kOpRecord, kStreamCaller, kEvL0);
        push(kOpWait, side, kEvL0);
    }
    for (uint32_t batch = 0; batch < n_batches; ++batch) {
        uint32_t par = overlap ? (batch & 1u) : 0u;               // buffer set of this batch
        uint32_t lane = 0;                                        // ... and, with lanes, the stream its launch goes to
        if (lanes) {
            par = s.set_next; s.set_next = (s.set_next + 1u) % (uint32_t)kSets;
            lane = s.lane_next; s.lane_next = (s.lane_next + 1u) % (uint32_t)kLanes;
        }
        // batch k reuses the sample buffer and hand-over queue of batch k - 2: wait until its tail is through
        if (overlap && batch >= 2) push(kOpWait, kStreamCaller, kEvResolved + par);
        for (uint32_t level = 0; level < n_levels; ++level) {
            uint32_t ls = level == 0 ? kStreamCaller : side;
            if (lanes) {
                // this lane's stream: behind the resolve that read the set's sample buffer last (kSets batches ago), and behind this
                // call's fills on the caller's stream, if any -- not behind the caller's stream as such
                ls = kStreamLane0 + lane;
                if (s.set_used[par]) push(kOpWait, ls, kEvSetFree + par);
                if (lanes_wait_pre && !lane_waited_pre[lane]) { push(kOpWait, ls, kEvPre); lane_waited_pre[lane] = true; }
                // ... and not before the previous lane's launch has been handed to the device: two launches that become ready at
                // the same moment would share the device from the start and run dry together
                const uint32_t prev_lane = (lane + (uint32_t)kLanes - 1u) % (uint32_t)kLanes;
                if (s.lane_used[prev_lane]) push(kOpWait, ls, kEvLaneBegun + prev_lane);
                push(kOpRecord, ls, kEvLaneBegun + lane);
                s.lane_used[lane] = 1;
            }
            const bool own = overlap && level > 0;       // continuation launch of an overlapped batch: its own queue
            Op l{};
            l.kind = kOpLaunch; l.stream = ls; l.set = par; l.lane = lane; l.level = level; l.own_queue = own ? 1u : 0u;
            l.ovf_par = lanes ? 0u : par;                // (with lanes nothing is ever handed over: the sets share one queue)
            l.batch = batch;
            l.grid = level > 0 ? j.cont_grid : j.regen ? j.regen_grid : j.grid;
            l.flags = (j.hand_off ? kLaunchHandOff : 0u) | (level == 0 ? kLaunchPrimary : 0u) | (lanes ? kLaunchLanes : 0u);
            if (level == 0 && j.regen) {
                l.flags |= kLaunchRegen | (lanes ? 0u : kLaunchStaticDeal);
                if (j.split) {
                    l.flags |= kLaunchSplit;
                    l.xchg_off = lanes ? (uint64_t)lane * stride : 0u;
                    l.xchg_len = j.xchg_need;
                }
                if (regen_core && regen_core < l.grid) {          // spare workgroups beyond the core
                    l.seq = ++s.lane_seq;
                    l.core = regen_core;
                    push(kOpPost, ls).seq = l.seq;                // before the launch is handed to the device
                }
            }
            if (j.hand_off && !j.capturing) s.counters_clean[par] = 0;       // until this batch's resolve has cleared them again
            if (profile) push(kOpRecord, ls, kEvPool, 2u * (ev0 + p.launches));
            p.ops.push_back(l);
            if (profile) push(kOpRecord, ls, kEvPool, 2u * (ev0 + p.launches) + 1u);
            if (level == 0) ++p.primary_launches;
            ++p.launches;
            if (overlap && level == 0) {     // the tail of this batch (side stream) starts when its level-0 launch is through
                push(kOpRecord, kStreamCaller, kEvL0 + par);
                push(kOpWait, side, kEvL0 + par);
            }
            if (lanes) {                     // the resolve (caller's stream) starts when the lane's launch is through
                push(kOpRecord, ls, kEvLaneDone + lane);
                if (!(faults & kFaultResolveBeforeLaunch)) push(kOpWait, kStreamCaller, kEvLaneDone + lane);
            }
        }
        Op& r = push(kOpResolve, side);
        r.set = par; r.batch = batch; r.zero_words = launch_words;
        if (launch_words && !j.capturing) s.counters_clean[par] = 1;
        if (overlap) push(kOpRecord, side, kEvResolved + par);
        if (lanes) { push(kOpRecord, kStreamCaller, kEvSetFree + par); s.set_used[par] = 1; }
    }
    if (overlap)        // the caller's stream is complete when the last resolve is (the side stream is in order)
        push(kOpWait, kStreamCaller, kEvResolved + ((n_batches - 1u) & 1u));
    push(kOpRecord, kStreamCaller, kEvEnd);
    if (profile) s.profiled = ev0 + p.launches;
    s.stats_pending = 1;
    if (j.capturing) s.captured_any = 1;
    return p;
}

// pt_sync: everything enqueued so far is complete -- the buffer sets and lanes start over; collected: the statistics were read and
// their words cleared (cleared_ok: the memset could be enqueued).
inline void on_sync(State& s, bool collected, bool cleared_ok) {
    s.set_next = 0; s.lane_next = 0;
    if (collected) { s.stats_clean = cleared_ok ? 1 : 0; s.stats_pending = 0; }
}
// pt_scene_upload: statistics of renders of the previous scene do not carry over
inline void on_scene(State& s) { s.stats_pending = 0; s.stats_clean = 0; s.captured_any = 0; }
// An operation of a render could not be enqueued.  The caller has waited for every stream (nothing is in flight any more); what
// ran of the render left counters and statistics in an unknown state.  The rotation starts over, nothing is known to be clean.
inline void on_failure(State& s, const State& planned) {
    State f;
    f.lane_seq = std::max(s.lane_seq, planned.lane_seq);   // the published launch number only ever grows (the cut-off plan may have published some)
    f.xchg_stride = s.xchg_stride;
    f.captured_any = s.captured_any;
    f.stream_work_since_lanes = 1;
    s = f;
}

}  // namespace ptsched
