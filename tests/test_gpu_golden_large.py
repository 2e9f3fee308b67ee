"""The BASELINE.json configurations at their FULL sizes, bit for bit against the CPU oracle.

The oracle prover needs 6 s .. 10 min per proof at these sizes, so it ran once in the build
container (tests/golden/make_golden_large.py) and left, per config, the Blake3 of the trace, every
commitment root (trace, quotient, each FRI round), the Blake3 of the opened values, final
polynomial, PoW witness and the Blake3 of all proof words in tests/golden/large_fixtures.json.
The GPU proof of the same trace must reproduce every one of them; the comparison runs in pipeline
order so a mismatch names the stage (reference uni-stark/src/prover.rs:25-119).

These are the shapes bench.py times: the 64-column strided NTT plan at n >= 2^20, the strided leaf
hash over 2^22 / 2^24 / 2^26 rows, the per-level Merkle launches with >= 2^19 parents.
"""
import os
import sys

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd.airs import FibonacciAir, SynthExtAir, SynthMulAir

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _digests import assert_matches_fixture, load_large  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from tapstark_amd.build import build

    build()
    return ts.default_context()


def device_trace(ctx, name, n):
    if name == "config2" or name.startswith("fib"):
        return ts.DeviceMatrix.fibonacci(ctx, 0, 1, n)
    if name in ("config3", "config4") or name.startswith("mul64"):
        return ts.DeviceMatrix.synth_mul(ctx, n, 64)
    return ts.DeviceMatrix.synth_ext(ctx, n, 163)


AIRS = {"config2": FibonacciAir, "config3": lambda: SynthMulAir(64), "config4": lambda: SynthMulAir(64),
        "config5": lambda: SynthExtAir(163), "fib_2p24_b2": FibonacciAir, "fib_2p25_b1": FibonacciAir,
        "fib_2p26_b1": FibonacciAir,
        "mul64_2p23_b1": lambda: SynthMulAir(64)}


# beyond BASELINE: n = 2^23 .. 2^26 (strided NTT passes of 11, 12, 13 and 14 stages: the generic plan), up
# to the 2^27-row LDE that is the field's limit
@pytest.mark.parametrize("name", ["config2", "config3", "config5", "config4", "fib_2p24_b2", "mul64_2p23_b1",
                                  "fib_2p25_b1", "fib_2p26_b1"])
def test_full_size_proof_equals_oracle_digests(ctx, orc, name):
    want = load_large(name)
    n = 1 << want["log_n"]
    cfg = (want["log_blowup"], want["num_queries"], want["proof_of_work_bits"])
    air = AIRS[name]()
    pis = np.array(want["public_values"], dtype=np.uint32)
    # the input first: the device-generated trace (what bench.py proves) is the oracle's trace
    host = device_trace(ctx, name, n).download()
    assert host.shape == (n, want["width"])
    assert orc.blake3(host.tobytes()).hex() == want["trace_blake3"], "device-generated trace differs"
    del host
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    proof = ts.prove(config, cair, ts.BfChallenger(), device_trace(ctx, name, n), pis)
    assert_matches_fixture(proof.words, want, f"{name} GPU proof")
    # and the interpreter path of the quotient at the same size (config 3 only: one more 3 ms proof)
    if name == "config3":
        os.environ["TS_NO_JIT"] = "1"
        try:
            cair2 = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
        finally:
            del os.environ["TS_NO_JIT"]
        assert not cair2.is_jit
        p2 = ts.prove(config, cair2, ts.BfChallenger(), device_trace(ctx, name, n), pis)
        assert (p2.words == proof.words).all()


@pytest.mark.parametrize("name,G,localq", [("config4", 8, False), ("config4", 8, True), ("config5", 8, False),
                                           ("config5", 8, True), ("config3", 4, False), ("config3", 2, True)])
def test_full_size_sharded_proof_equals_oracle_digests(ctx, orc, name, G, localq):
    """The same fixtures for ONE proof sharded over G ranks (threads on the box's one GPU, native
    in-process communicator): config 4 as BASELINE.json writes it (8 ranks,
    two cosets each), config 5's "on 8 x MI355X" half (163 columns over 8 ranks), and the headline
    config over the 4 ranks its log_blowup allows -- with the chunk broadcast and with every rank
    computing the quotient on its own cosets (local_quotient; config 3 then over 2 ranks: the
    quotient degree 2 needs two cosets per rank)."""
    import threading

    from tapstark_amd.comm import LocalCommGroup

    want = load_large(name)
    n = 1 << want["log_n"]
    cfg = (want["log_blowup"], want["num_queries"], want["proof_of_work_bits"])
    tape = ts.air_tape(AIRS[name](), 0)
    group = LocalCommGroup(G)
    proofs, errors = [None] * G, [None] * G

    def rank_main(r):
        try:
            c = ts.Context(0)
            conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c))
            p = ts.prove_sharded(conf, ts.CompiledAir(c, tape), ts.BfChallenger(), device_trace(c, name, n),
                                 [], group.comm(r), trace_replicated=True, local_quotient=localq)
            proofs[r] = p.words
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    for r in range(G):
        assert errors[r] is None, f"rank {r}: {errors[r]!r}"
    assert_matches_fixture(proofs[0], want, f"{name} sharded over {G}{' (local quotient)' if localq else ''}, rank 0")
    for r in range(1, G):
        assert (proofs[r] == proofs[0]).all(), f"rank {r} holds another proof"
