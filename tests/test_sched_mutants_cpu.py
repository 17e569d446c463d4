"""Mutation check of the scheduler's simulator (tests/sched_sim.py): does it notice when the planner is wrong?

Each mutant is csrc/pt_sched.h with ONE ordering rule removed or bent (a wait dropped, a fill skipped, the lanes switched on
where they must be off ...), compiled with g++ into a library of its own (pt_sched.cpp needs nothing else) and driven through
the same random sequences as the real planner.  Every mutant must be caught; the unmodified header must pass.  (The two bugs
round 4 actually had are in test_sched_cpu.py, behind the planner's own `faults` switch.)"""
import ctypes as C
import os
import random
import subprocess

import pytest

from pathtrace_amd._lib import PtSchedJob, PtSchedOp
from tests import sched_sim as S
from tests import test_sched_cpu as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pathtrace_amd", "csrc")

STUB = r'''
#include <cstdarg>
#include <cstdio>
static thread_local char g_err[512];
int pt_internal_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return code; }
extern "C" const char* pt_last_error(void) { return g_err; }
'''

# (name, text in pt_sched.h, replacement, invariant letters one of which must fire)
MUTANTS = [
    ("launch does not wait for the resolve that read its buffer set last",
     "if (s.set_used[par]) push(kOpWait, ls, kEvSetFree + par);", ";", "RC"),
    ("resolve does not wait for its lane's launch",
     "if (!(faults & kFaultResolveBeforeLaunch)) push(kOpWait, kStreamCaller, kEvLaneDone + lane);", ";", "RD"),
    ("lanes do not wait for the fills on the caller's stream",
     "if (lanes_wait_pre && !lane_waited_pre[lane]) { push(kOpWait, ls, kEvPre); lane_waited_pre[lane] = true; }", ";", "R"),
    ("the resolve does not clear the launch counters",
     "r.zero_words = launch_words;", "r.zero_words = 0;", "C"),
    ("the caller's stream does not wait for the side stream's last resolve",
     "push(kOpWait, kStreamCaller, kEvResolved + ((n_batches - 1u) & 1u));", ";", "DR"),
    ("batch k of an overlapped render does not wait for the tail of batch k - 2",
     "if (overlap && batch >= 2) push(kOpWait, kStreamCaller, kEvResolved + par);", ";", "R"),
    ("dirty launch counters are not cleared",
     "if (launch_words && (j.capturing || !s.counters_clean[par])) {", "if (false) {", "C"),
    ("dirty statistics are not cleared",
     "if (!j.capturing && (s.captured_any || !s.stats_clean)) push(kOpMemsetStats, kStreamCaller);", ";", "T"),
    ("profiled renders take the lanes",
     "return kLaneOverlap && j.regen && j.regen_export <= 1u && !j.profile && !j.in_order && !j.capturing;",
     "return kLaneOverlap && j.regen && j.regen_export <= 1u && !j.in_order && !j.capturing;", "LP"),
    ("captured renders take the lanes",
     "return kLaneOverlap && j.regen && j.regen_export <= 1u && !j.profile && !j.in_order && !j.capturing;",
     "return kLaneOverlap && j.regen && j.regen_export <= 1u && !j.profile && !j.in_order;", "LRCE"),
    ("a lanes launch deals chunks statically",
     "l.flags |= kLaunchRegen | (lanes ? 0u : kLaunchStaticDeal);", "l.flags |= kLaunchRegen | kLaunchStaticDeal;", "L"),
    ("the launch number is not advanced",
     "l.seq = ++s.lane_seq;", "l.seq = s.lane_seq;", "P"),
    ("the exchange stride grows under running launches",
     "push(kOpHostSync, kStreamCaller);", ";", "X"),
    ("a render without lanes is forgotten",
     "s.stream_work_since_lanes = lanes ? 0 : 1;", "s.stream_work_since_lanes = 0;", "RC"),
    ("the tail of an overlapped batch does not wait for its level-0 launch",
     "push(kOpWait, side, kEvL0 + par);", ";", "R"),
    ("every lanes launch goes to lane 0's region",
     "l.xchg_off = lanes ? (uint64_t)lane * stride : 0u;", "l.xchg_off = 0u;", "X"),
    ("the resolve reads buffer set 0 whatever set the launch wrote",
     "r.set = par; r.batch = batch; r.zero_words = launch_words;", "r.set = 0; r.batch = batch; r.zero_words = launch_words;", "SRC"),
    ("a captured render trusts the host's flags",
     "if (launch_words && (j.capturing || !s.counters_clean[par])) {", "if (launch_words && !s.counters_clean[par]) {", "C"),
]


def build(tmp, text, name):
    d = os.path.join(tmp, name)
    os.makedirs(d, exist_ok=True)
    os.makedirs(os.path.join(d, "a", "b"), exist_ok=True)
    with open(os.path.join(d, "a", "b", "pt_sched.h"), "w") as f:
        f.write(text)
    src = open(os.path.join(CSRC, "pt_sched.cpp")).read().replace('#include "../../include/pathtrace_amd.h"',
                                                                   '#include "%s"' % os.path.join(ROOT, "include", "pathtrace_amd.h"))
    with open(os.path.join(d, "a", "b", "pt_sched.cpp"), "w") as f:
        f.write(src + STUB)
    so = os.path.join(d, "libsched.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(d, "a", "b", "pt_sched.cpp")], check=True)
    h = C.CDLL(so)
    h.pt_debug_sched_create.argtypes = [C.POINTER(C.c_void_p)]
    h.pt_debug_sched_destroy.argtypes = [C.c_void_p]; h.pt_debug_sched_destroy.restype = None
    h.pt_debug_sched_render.argtypes = [C.c_void_p, C.POINTER(PtSchedJob), C.c_uint32, C.c_uint32, C.POINTER(PtSchedOp), C.c_uint32,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    h.pt_debug_sched_sync.argtypes = [C.c_void_p, C.c_uint32]
    h.pt_last_error.restype = C.c_char_p
    return h


def hunt(lib, monkeypatch):
    """the random sequences of test_sched_cpu.py against `lib`; returns the first violation's letter or None"""
    class Bound(T.Sched):
        def __init__(self):
            self.lib = lib
            self.h = C.c_void_p()
            assert lib.pt_debug_sched_create(C.byref(self.h)) == 0
            self.buf = (PtSchedOp * T.CAP)()
    monkeypatch.setattr(T, "Sched", Bound)
    for seed in range(600):
        try:
            T.drive(seed, n_renders=random.Random(seed).randrange(2, 14))
        except S.Violation as e:
            return str(e)[0]
    for seed in range(60):
        try:
            T.drive(20_000 + seed, n_renders=40, with_failures=False)
        except S.Violation as e:
            return str(e)[0]
    return None


def test_every_mutant_of_the_planner_is_caught(tmp_path, monkeypatch):
    header = open(os.path.join(CSRC, "pt_sched.h")).read()
    assert hunt(build(str(tmp_path), header, "original"), monkeypatch) is None        # the harness itself: the real planner passes
    survivors = []
    for k, (name, old, new, letters) in enumerate(MUTANTS):
        assert header.count(old) == 1, f"mutant {k} ({name}): the line it bends is not in pt_sched.h any more"
        got = hunt(build(str(tmp_path), header.replace(old, new), f"m{k}"), monkeypatch)
        if got is None or got not in letters:
            survivors.append((name, got))
    assert not survivors, survivors
