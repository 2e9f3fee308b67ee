"""Stage-localising digests of one TSPF v1 proof: what tests/golden/large_fixtures.json stores per
BASELINE config (made by tests/golden/make_golden_large.py from the CPU oracle) and what the GPU
tests and bench.py recompute from a GPU proof.  Test infrastructure (uses the oracle's Blake3)."""
import json
import os

import numpy as np

LARGE_FIXTURES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "large_fixtures.json")


def hexd(words) -> str:
    return np.ascontiguousarray(words, dtype=np.uint32).tobytes().hex()


def digest_record(proof_words: np.ndarray) -> dict:
    import tapstark_amd as ts
    from oracle import oracle_py as orc

    pf = ts.Proof.parse(np.ascontiguousarray(proof_words, dtype=np.uint32))
    opened = np.concatenate([pf.trace_local.reshape(-1), pf.trace_next.reshape(-1),
                             pf.quotient_chunks.reshape(-1)])
    return {
        "proof_words": int(len(pf.words)),
        "proof_blake3": orc.blake3(pf.words.tobytes()).hex(),
        "trace_commit": hexd(pf.trace_commit),
        "quotient_commit": hexd(pf.quotient_commit),
        "opened_values_blake3": orc.blake3(np.ascontiguousarray(opened).tobytes()).hex(),
        "commit_phase_commits": [hexd(c) for c in pf.commit_phase_commits],
        "final_poly": [int(x) for x in pf.final_poly],
        "pow_witness": int(pf.pow_witness),
    }


def load_large(name: str) -> dict:
    return json.load(open(LARGE_FIXTURES))[name]


def assert_matches_fixture(proof_words, want: dict, what: str = "GPU proof"):
    """Compares in pipeline order, so the first failing assertion names the first stage that differs
    (reference uni-stark/src/prover.rs:25-119: trace commit :53, quotient commit :82-83, opened
    values :94-104, FRI commit phase fri/src/prover.rs:93-141, final poly, PoW, then everything)."""
    got = digest_record(proof_words)
    assert got["trace_commit"] == want["trace_commit"], f"{what}: trace commitment (LDE / leaf hash / Merkle) differs"
    assert got["quotient_commit"] == want["quotient_commit"], f"{what}: quotient commitment differs"
    assert got["opened_values_blake3"] == want["opened_values_blake3"], f"{what}: opened values differ"
    for r, (a, b) in enumerate(zip(got["commit_phase_commits"], want["commit_phase_commits"])):
        assert a == b, f"{what}: FRI commit-phase root of round {r} differs"
    assert len(got["commit_phase_commits"]) == len(want["commit_phase_commits"])
    assert got["final_poly"] == want["final_poly"], f"{what}: final polynomial differs"
    assert got["pow_witness"] == want["pow_witness"], f"{what}: PoW witness differs"
    assert got["proof_words"] == want["proof_words"]
    assert got["proof_blake3"] == want["proof_blake3"], f"{what}: query phase (rows / paths) differs"
