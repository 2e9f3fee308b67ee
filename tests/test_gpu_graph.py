"""TS_FRI_GRAPH: the FRI commit phase (fri/src/prover.rs:93-141) captured into a hipGraph.

Round 3 left a failure on record (TS_ERR_OOM at the first config-2-shaped proof after config-3-shaped
ones on one context: the capture reached hipMalloc).  The pool now serves a capture from reserved
blocks only, and an allocation that still misses ends the capture and re-runs the phase eagerly."""
import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd.airs import (FibonacciAir, SynthMulAir, fibonacci_public_values,
                               generate_fibonacci_trace, generate_synth_mul_trace)

pytestmark = pytest.mark.gpu


def cases(log_n):
    fib = generate_fibonacci_trace(0, 1, 1 << log_n)
    return [("mul64", SynthMulAir(64), generate_synth_mul_trace(1 << log_n), np.zeros(0, dtype=np.uint32)),
            ("fib", FibonacciAir(), fib, fibonacci_public_values(fib))]


@pytest.mark.parametrize("knob,log_n", [("1", 16), ("2", 16), ("3", 16), ("1", 20)])
def test_graph_knob_same_context_two_shapes(monkeypatch, knob, log_n):
    from tapstark_amd.build import build

    build()
    monkeypatch.delenv("TS_FRI_GRAPH", raising=False)
    ref_ctx = ts.Context(0)
    cfg = (2, 28, 8)
    want = {}
    for name, air, trace, pis in cases(log_n):
        conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ref_ctx))
        want[name] = ts.prove(conf, air, ts.BfChallenger(), trace.copy(), pis).words
    assert ref_ctx.graph_stats()["replays"] == 0
    # knob on: config-3-shaped proofs, then config-2-shaped ones, on ONE context (same log_max_height
    # and input count -- the two shapes round 3's warm-up key could not tell apart), then back
    monkeypatch.setenv("TS_FRI_GRAPH", knob)
    ctx = ts.Context(0)
    conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cs = cases(log_n)
    order = [cs[0]] * 3 + [cs[1]] * 3 + [cs[0]] * 2 + [cs[1]]
    for name, air, trace, pis in order:
        ch = ts.BfChallenger()
        got = ts.prove(conf, air, ch, trace.copy(), pis).words
        assert len(got) == len(want[name]) and (got == want[name]).all(), f"{name}: proof differs with the graph"
    st = ctx.graph_stats()
    # both AIRs open one vector of the same height: one shape, 1 recording proof, 8 graph attempts
    assert st["shapes"] == 1
    if knob == "1":
        assert st["replays"] == 8 and st["fallbacks"] == 0, st
    elif knob == "2":  # no reservation: parked frees starve the capture, the eager fall-back must take over
        assert st["replays"] + st["fallbacks"] == 8 and st["fallbacks"] >= 1, st
    else:  # a good capture whose replay "fails" (instantiate / launch error): the same fall-back, every time
        assert st["replays"] == 0 and st["fallbacks"] == 8, st
    assert st["reserve_failures"] == 0
    # another blowup on the same context: a new shape, recorded first, then replayed
    conf3 = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(3, 9, 8), ctx))
    name, air, trace, pis = cases(10)[0]
    monkeypatch.delenv("TS_FRI_GRAPH")
    w3 = ts.prove(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(3, 9, 8), ref_ctx)), air, ts.BfChallenger(),
                  trace.copy(), pis).words
    monkeypatch.setenv("TS_FRI_GRAPH", knob)
    for _ in range(3):
        assert (ts.prove(conf3, air, ts.BfChallenger(), trace.copy(), pis).words == w3).all()
    assert ctx.graph_stats()["shapes"] == 2
