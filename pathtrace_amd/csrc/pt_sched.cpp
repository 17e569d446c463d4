// pt_sched.cpp -- pt_debug_sched_*: the launch planner of pt_sched.h on a scheduling state of its own (host only, no HIP).
// What pt_api.cpp's render_impl does with a context's state -- plan on a copy, execute, commit; recover on a failure -- is
// done here with the "execute" step left to the caller: tests/test_sched_cpu.py simulates the streams.
#include <cstring>
#include <new>

#include "../../include/pathtrace_amd.h"
#include "pt_sched.h"

int pt_internal_fail(int code, const char* fmt, ...);

struct PtSched {
    ptsched::State state;
};

static_assert(sizeof(PtSchedOp) == sizeof(ptsched::Op), "PtSchedOp mirrors ptsched::Op");
static_assert(sizeof(PtSchedJob) == 14 * sizeof(uint32_t) + sizeof(uint64_t), "PtSchedJob layout");

extern "C" {

int pt_debug_sched_create(PtSched** out) {
    if (!out) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_sched_create: out is null");
    *out = new (std::nothrow) PtSched();
    return *out ? PT_OK : pt_internal_fail(PT_ERR_OOM, "pt_debug_sched_create: out of memory");
}

void pt_debug_sched_destroy(PtSched* s) { delete s; }

int pt_debug_sched_render(PtSched* s, const PtSchedJob* job, uint32_t faults, uint32_t fail_after, PtSchedOp* ops, uint32_t cap,
                          uint32_t* n_ops, uint32_t* lanes) {
    if (!s || !job || !n_ops || (!ops && cap)) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_sched_render: null argument");
    ptsched::Job j;
    j.n_batches = job->n_batches; j.regen = job->regen; j.split = job->regen ? job->split : 0u; j.hand_off = job->hand_off;
    j.regen_export = job->regen_export; j.profile = job->profile; j.in_order = job->in_order; j.capturing = job->capturing;
    j.grid = job->grid; j.regen_grid = job->regen_grid; j.cont_grid = job->cont_grid; j.regen_capacity = job->regen_capacity;
    j.fixed_grid = job->fixed_grid; j.counter_words = job->counter_words; j.xchg_need = j.split ? job->xchg_need : 0u;
    if (j.regen && !j.hand_off) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_sched_render: a regenerating launch uses the chunk counters (hand_off = 1)");
    ptsched::State next = s->state;                        // render_impl: plan on a copy ...
    ptsched::Plan p = ptsched::plan(next, j, faults);
    if (lanes) *lanes = p.lanes ? 1u : 0u;
    size_t n = p.ops.size();
    const bool failed = fail_after < n;
    if (failed) {
        // ... the operations before the failing one were enqueued; the recovery waits for every stream and starts the state over
        n = fail_after;
        ptsched::Op sync{};
        sync.kind = ptsched::kOpHostSync;
        p.ops.resize(n);
        p.ops.push_back(sync);
        n += 1;
        ptsched::on_failure(s->state, next);
    } else {
        s->state = next;                                    // ... and commit it after the last operation
    }
    *n_ops = (uint32_t)n;
    if (n > cap) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_sched_render: %zu operations, room for %u", n, cap);
    if (n) std::memcpy(ops, p.ops.data(), n * sizeof(PtSchedOp));
    return failed ? PT_ERR_HIP : PT_OK;
}

int pt_debug_sched_sync(PtSched* s, uint32_t collect) {
    if (!s) return pt_internal_fail(PT_ERR_INVALID_ARG, "pt_debug_sched_sync: null argument");
    const bool c = collect != 0 && s->state.stats_pending != 0;
    ptsched::on_sync(s->state, c, true);
    return PT_OK;
}

}  // extern "C"
