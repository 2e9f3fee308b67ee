"""CPU side of the full-size fixtures: tests/golden/large_fixtures.json must be what the committed
oracle produces (so the GPU tests compare against the oracle, not against a stale file).  Configs 2
and 3 are regenerated here (6 s and 22 s of oracle time on 8 cores); configs 4 and 5 take minutes
and 40 GiB and are checked for presence and shape only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _digests import assert_matches_fixture, load_large  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd.airs import (FibonacciAir, SynthMulAir, fibonacci_public_values,  # noqa: E402
                               generate_fibonacci_trace, generate_synth_mul_trace)


def test_fixture_shapes():
    for name, (log_n, w, b, q) in {"config2": (20, 2, 2, 28), "config3": (20, 64, 2, 28),
                                   "config4": (22, 64, 4, 16), "config5": (20, 163, 4, 16)}.items():
        f = load_large(name)
        assert (f["log_n"], f["width"], f["log_blowup"], f["num_queries"]) == (log_n, w, b, q)
        assert len(f["commit_phase_commits"]) == log_n
        assert len(f["betas"]) == log_n and len(f["query_indices"]) == q
        for key in ("trace_blake3", "proof_blake3", "trace_commit", "quotient_commit", "opened_values_blake3"):
            assert len(f[key]) == 64


def test_config2_fixture_is_the_oracles(orc):
    want = load_large("config2")
    trace = generate_fibonacci_trace(0, 1, 1 << 20)
    pis = fibonacci_public_values(trace)
    assert [int(x) for x in pis] == want["public_values"]
    assert orc.blake3(trace.tobytes()).hex() == want["trace_blake3"]
    proof = orc.prove(orc.FriConfig(2, 28, 8), ts.air_tape(FibonacciAir(), 3), trace, pis)
    assert_matches_fixture(proof, want, "oracle proof")


def test_config3_fixture_is_the_oracles(orc):
    want = load_large("config3")
    trace = generate_synth_mul_trace(1 << 20)
    assert orc.blake3(trace.tobytes()).hex() == want["trace_blake3"]
    proof = orc.prove(orc.FriConfig(2, 28, 8), ts.air_tape(SynthMulAir(64), 0), trace, np.zeros(0, dtype=np.uint32))
    assert_matches_fixture(proof, want, "oracle proof")
    tr = orc.last_transcript()
    assert [int(x) for x in tr["zeta"]] == want["zeta"]
    assert [int(x) for x in tr["indices"]] == want["query_indices"]
