"""GPU parity tests: every stage of the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs (bit-exact: all arithmetic is integer modular), then whole proofs
byte-for-byte, then size-independent properties at the BASELINE sizes."""
import json
import os

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd.airs import (FibonacciAir, SynthExtAir, SynthMulAir, fibonacci_public_values,
                               generate_fibonacci_trace, generate_synth_ext_trace,
                               generate_synth_mul_trace, splitmix64_stream)

pytestmark = pytest.mark.gpu
P = 0x78000001
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rand_mat(seed, h, w):
    return splitmix64_stream(seed, h * w).reshape(h, w)


@pytest.fixture(scope="module")
def ctx():
    from tapstark_amd.build import build

    build()
    return ts.default_context()


def bitrev_perm(bits):
    idx = np.arange(1 << bits, dtype=np.uint64)
    out = np.zeros_like(idx)
    for b in range(bits):
        out |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(bits - 1 - b)
    return out.astype(np.int64)


# ------------------------------------------------------------------ commit: LDE + Merkle
@pytest.mark.parametrize("log_n,w,log_blowup", [(0, 1, 1), (1, 2, 2), (3, 2, 2), (5, 7, 1), (6, 64, 2),
                                                (10, 64, 2), (12, 5, 2), (13, 3, 2), (14, 64, 2),
                                                (16, 2, 3), (11, 163, 2), (22, 1, 1),
                                                # 2^21 / 2^22: 8192- / 16384-element chunks (radix-32 rounds)
                                                (21, 2, 2), (22, 2, 2), (21, 1, 4),
                                                # 2^23 .. 2^26 rows: strided passes of 11 .. 14 stages (generic plan);
                                                # (26, 1, 1) is the largest LDE the field allows (2^27 rows)
                                                (23, 3, 1), (24, 1, 2), (25, 2, 1), (26, 1, 1)])
def test_commit_lde_and_merkle(ctx, orc, log_n, w, log_blowup):
    if log_n >= 24 and not os.environ.get("TS_BIG_TESTS"):
        # 15-30 s of oracle hashing each; run with TS_BIG_TESTS=1 (round 4: all passed).
        # The same LDE plans (2^24, 2^25, 2^26 rows) are covered end to end by the whole-proof digests of
        # fib_2p24_b2 / fib_2p25_b1 / fib_2p26_b1 in test_gpu_golden_large.py, which cost a second each.
        pytest.skip("2^24+ rows against every oracle digest level: TS_BIG_TESTS=1")
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(log_blowup, 4, 8), ctx)
    m = rand_mat(17 + log_n, 1 << log_n, w)
    # a second, non-trivial domain shift where the field has one and the case is not a big one (the
    # 2^24+ cases take tens of seconds of oracle hashing each)
    for shift in ((1, 31 * pow(0x1A427A41, 1 << (27 - (log_n + 1)), P) % P) if log_n < 24 else (1,)):
        root, data = pcs.commit([((log_n, shift), m.copy())])
        want = orc.commit_lde(m, shift, log_blowup)
        got = data.lde(0, w)
        assert got.shape == want.shape
        assert (got == want).all(), f"LDE mismatch log_n={log_n} w={w} shift={shift}: " \
                                    f"{int((got != want).sum())} of {got.size} differ"
        om = orc.OracleMmcs([want])
        assert (data.digests(0) == om.layer(0)).all(), "leaf digests differ"
        for lvl in range(1, data.log_height + 1):
            assert (data.digests(lvl) == om.layer(lvl)).all(), f"digest level {lvl} differs"
        assert (root == om.root).all()
        for idx in {0, (1 << data.log_height) - 1, (5 * 977) % (1 << data.log_height)}:
            rows, path = data.open_batch(idx, w)
            orows, opath = om.open(idx)
            assert (rows == orows).all() and (path == opath).all()
            assert om.verify(idx, rows, path, root)


def test_commit_batch_of_matrices(ctx, orc):
    # a batch of several matrices (the quotient-chunk commit shape): leaf = row0 || row1 || ...
    log_n, b = 7, 2
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 4, 8), ctx)
    mats = [rand_mat(40 + i, 1 << log_n, 4) for i in range(4)]
    g = pow(0x1A427A41, 1 << (27 - (log_n + 2)), P)
    shifts = [31 * pow(g, c, P) % P for c in range(4)]
    root, data = pcs.commit([((log_n, s), m.copy()) for s, m in zip(shifts, mats)])
    ldes = [orc.commit_lde(m, s, b) for s, m in zip(shifts, mats)]
    for i in range(4):
        assert (data.lde(i, 4) == ldes[i]).all()
    om = orc.OracleMmcs(ldes)
    assert (root == om.root).all()
    rows, path = data.open_batch(77, 16)
    assert om.verify(77, rows, path, root)


# ------------------------------------------------------------------ quotient
AIRS = [
    ("fib", lambda n: (FibonacciAir(), generate_fibonacci_trace(0, 1, n)), True),
    ("mul64", lambda n: (SynthMulAir(64), generate_synth_mul_trace(n)), False),
    ("mul7", lambda n: (SynthMulAir(7), generate_synth_mul_trace(n, 7)), False),
    ("ext25", lambda n: (SynthExtAir(25), generate_synth_ext_trace(n, 25)), False),
]


@pytest.mark.parametrize("name,make,has_pis", AIRS, ids=[a[0] for a in AIRS])
@pytest.mark.parametrize("log_n", [3, 8, 13])
@pytest.mark.parametrize("jit", [True, False], ids=["jit", "interp"])
def test_quotient_chunks(ctx, orc, monkeypatch, name, make, has_pis, log_n, jit):
    # both quotient paths run on the GPU: the hiprtc-specialised kernel and the tape interpreter
    if not jit:
        monkeypatch.setenv("TS_NO_JIT", "1")
    b = 2
    air, trace = make(1 << log_n)
    pis = fibonacci_public_values(trace) if has_pis else np.zeros(0, dtype=np.uint32)
    tape = ts.air_tape(air, len(pis))
    cair = ts.CompiledAir(ctx, tape)
    assert cair.is_jit == jit, "hiprtc specialisation expected on the GPU box"
    assert cair.log_quotient_degree == orc.log_quotient_degree(tape)
    assert cair.max_constraint_degree == orc.max_constraint_degree(tape)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 4, 8), ctx)
    _, data = pcs.commit([((log_n, 1), trace.copy())])
    alpha = rand_mat(5, 1, 4)[0]
    chunks = pcs.quotient_chunks(data, cair, pis, alpha)
    lde = orc.commit_lde(trace, 1, b)
    want = orc.split_quotient(orc.quotient_values(tape, lde, log_n, b, pis, alpha), log_n,
                              cair.log_quotient_degree)
    assert len(chunks) == want.shape[0]
    for c, ch in enumerate(chunks):
        got = ch.download()
        assert (got == want[c]).all(), f"chunk {c}: {int((got != want[c]).sum())} words differ"


# ------------------------------------------------------------------ open / reduce
@pytest.mark.parametrize("log_n,w,qd", [(3, 2, 1), (6, 5, 2), (10, 64, 2), (13, 9, 4)])
def test_open_reduce(ctx, orc, log_n, w, qd):
    b = 2
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 4, 8), ctx)
    trace = rand_mat(60, 1 << log_n, w)
    chunks = [rand_mat(61 + c, 1 << log_n, 4) for c in range(qd)]
    lqd = qd.bit_length() - 1
    g = pow(0x1A427A41, 1 << (27 - (log_n + lqd)), P) if log_n + lqd else 1
    shifts = [31 * pow(g, c, P) % P for c in range(qd)]
    _, tdata = pcs.commit([((log_n, 1), trace.copy())])
    _, qdata = pcs.commit([((log_n, s), m.copy()) for s, m in zip(shifts, chunks)])
    zeta, alpha = rand_mat(70, 1, 4)[0], rand_mat(71, 1, 4)[0]
    opened, ro = pcs.open_reduce(tdata, qdata, w, zeta, alpha)
    tl = orc.commit_lde(trace, 1, b)
    cl = [orc.commit_lde(m, s, b) for s, m in zip(shifts, chunks)]
    want_opened, want_ro = orc.open_reduce(tl, cl, log_n, b, zeta, alpha)
    assert (opened == want_opened).all(), "opened values differ"
    assert (ro == want_ro).all(), f"reduced openings differ in {int((ro != want_ro).any(axis=1).sum())} rows"


# ------------------------------------------------------------------ FRI fold
@pytest.mark.parametrize("log_h", [0, 1, 4, 11, 16])
def test_fri_fold(ctx, orc, log_h):
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(1, 4, 8), ctx)
    vec = rand_mat(80 + log_h, 2 << log_h, 4)
    beta = rand_mat(81, 1, 4)[0]
    assert (pcs.fold_matrix(vec, beta) == orc.fold_matrix(vec, beta)).all()


# ------------------------------------------------------------------ whole proofs
PROOF_CASES = [
    ("fib8_q28", lambda: (FibonacciAir(), generate_fibonacci_trace(0, 1, 8)), True, (2, 28, 8)),
    ("fib8_q16", lambda: (FibonacciAir(), generate_fibonacci_trace(0, 1, 8)), True, (2, 16, 8)),
    ("fib2", lambda: (FibonacciAir(), generate_fibonacci_trace(0, 1, 2)), True, (2, 5, 8)),
    ("fib_2p12_b1", lambda: (FibonacciAir(), generate_fibonacci_trace(0, 1, 1 << 12)), True, (1, 9, 8)),
    ("fib_2p14", lambda: (FibonacciAir(), generate_fibonacci_trace(3, 5, 1 << 14)), True, (2, 28, 8)),
    ("mul64_2p10", lambda: (SynthMulAir(64), generate_synth_mul_trace(1 << 10)), False, (2, 28, 8)),
    ("mul64_2p13_b4", lambda: (SynthMulAir(64), generate_synth_mul_trace(1 << 13)), False, (4, 16, 8)),
    ("mul7_2p6_b3", lambda: (SynthMulAir(7), generate_synth_mul_trace(1 << 6, 7)), False, (3, 7, 4)),
    ("ext163_2p8", lambda: (SynthExtAir(163), generate_synth_ext_trace(1 << 8, 163)), False, (2, 16, 8)),
    # BASELINE config 5's FRI parameters (log_blowup 4, 16 queries; README.md:91,101) at sizes the
    # oracle prover finishes in seconds
    ("ext163_2p8_b4", lambda: (SynthExtAir(163), generate_synth_ext_trace(1 << 8, 163)), False, (4, 16, 8)),
    ("ext163_2p11_b4", lambda: (SynthExtAir(163), generate_synth_ext_trace(1 << 11, 163)), False, (4, 16, 8)),
    # a constraint of degree 33: quotient_degree 32 (was refused above 16), needs log_blowup >= 5
    ("deg33_2p6_b5", lambda: (__import__("tapstark_amd").airs.HighDegreeAir(33),
                              __import__("tapstark_amd").airs.generate_high_degree_trace(1 << 6)), False, (5, 6, 4)),
    ("deg50_2p4_b6", lambda: (__import__("tapstark_amd").airs.HighDegreeAir(50),
                              __import__("tapstark_amd").airs.generate_high_degree_trace(1 << 4)), False, (6, 3, 4)),
    # whole proofs through the 8192- / 16384-element NTT chunk plans (2^21 / 2^22 rows), narrow so that
    # the oracle prover finishes in seconds
    ("mul7_2p21_b2", lambda: (SynthMulAir(7), generate_synth_mul_trace(1 << 21, 7)), False, (2, 7, 8)),
    ("fib_2p22_b1", lambda: (FibonacciAir(), generate_fibonacci_trace(1, 2, 1 << 22)), True, (1, 5, 8)),
]


@pytest.mark.parametrize("name,make,has_pis,cfg", PROOF_CASES, ids=[c[0] for c in PROOF_CASES])
def test_prove_bit_identical_to_oracle(ctx, orc, name, make, has_pis, cfg):
    air, trace = make()
    pis = fibonacci_public_values(trace) if has_pis else np.zeros(0, dtype=np.uint32)
    tape = ts.air_tape(air, len(pis))
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    challenger = ts.BfChallenger()
    proof = ts.prove(config, air, challenger, trace, pis)
    ocfg = orc.FriConfig(*cfg)
    ochal = orc.OracleChallenger()
    want = orc.prove(ocfg, tape, trace, pis, ochal)
    assert orc.verify(ocfg, tape, proof.words, pis) == 0, "oracle verifier rejects the GPU proof"
    # the reference's own test shape (uni-stark/tests/fib_air.rs:144-148): prove, then verify with
    # a fresh challenger -- here with the product's native verifier
    ts.verify(config, air, ts.BfChallenger(), proof, pis)
    assert len(proof.words) == len(want)
    assert (proof.words == want).all(), f"{int((proof.words != want).sum())} proof words differ"
    # the caller's challenger must end in the same transcript state (prover.rs takes &mut)
    st = challenger.state()
    assert st[:16].tolist() == list(ochal.c.state)
    assert challenger.sample_bits(20) == ochal.sample_bits(20)
    assert proof.degree_bits == trace.shape[0].bit_length() - 1


def test_fib8_golden_fixture(ctx, orc):
    golden = json.load(open(os.path.join(GOLDEN, "oracle_fixtures.json")))
    trace = generate_fibonacci_trace(0, 1, 8)
    pis = fibonacci_public_values(trace)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 28, 8), ctx))
    proof = ts.prove(config, FibonacciAir(), ts.BfChallenger(), trace, pis)
    assert orc.blake3(proof.words.tobytes()).hex() == golden["fib8_q28_proof_blake3"]
    assert proof.trace_commit.tolist() == golden["fib8_trace_root"]
    assert proof.pow_witness == golden["fib8_q28_pow_witness"]


def test_synthmul_golden_fixture(ctx, orc):
    golden = json.load(open(os.path.join(GOLDEN, "oracle_fixtures.json")))
    trace = generate_synth_mul_trace(1 << 10)
    assert orc.blake3(trace.tobytes()).hex() == golden["synthmul64_2pow10_trace_blake3"]
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 28, 8), ctx))
    proof = ts.prove(config, SynthMulAir(64), ts.BfChallenger(), trace, [])
    assert orc.blake3(proof.words.tobytes()).hex() == golden["synthmul64_2pow10_proof_blake3"]


def test_error_behaviour(ctx):
    from tapstark_amd._lib import TsError

    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 8), ctx))
    with pytest.raises(TsError):  # width mismatch
        ts.prove(config, FibonacciAir(), ts.BfChallenger(), rand_mat(1, 8, 3), [0, 1, 21])
    with pytest.raises(TsError):  # not a power of two
        ts.DeviceMatrix.upload(ctx, rand_mat(1, 6, 2))
    with pytest.raises(TsError):  # wrong number of public values
        ts.prove(config, ts.CompiledAir(ctx, ts.air_tape(FibonacciAir(), 3)), ts.BfChallenger(),
                 generate_fibonacci_trace(0, 1, 8), [0, 1])
    # log_quotient_degree (1) > log_blowup (... must be >= ): two_adic_pcs.rs:256 assert
    low = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(1, 4, 8), ctx))
    ts.prove(low, SynthMulAir(7), ts.BfChallenger(), generate_synth_mul_trace(16, 7), [])  # qd=2 fits b=1
    with pytest.raises(TsError):  # malformed tape
        ts.CompiledAir(ctx, np.array([1, 2, 3, 4, 5, 6], dtype=np.uint32))


def test_invalid_trace_is_caught_by_the_verifier(ctx, orc):
    """A release build of the reference proves an invalid trace without complaint (its
    check_constraints is debug-only, prover.rs:40-41; with quotient_degree 1 the quotient always
    interpolates); the verifier's out-of-domain check (verifier.rs:157) must then reject."""
    bad = generate_fibonacci_trace(0, 1, 64)
    bad[7, 0] += 1
    pis = fibonacci_public_values(bad)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 6, 8), ctx))
    proof = ts.prove(config, FibonacciAir(), ts.BfChallenger(), bad, pis)
    tape = ts.air_tape(FibonacciAir(), 3)
    assert orc.verify(orc.FriConfig(2, 6, 8), tape, proof.words, pis) == 7  # OodEvaluationMismatch
    # same with quotient_degree 2: every chunk is still an honest low-degree extension of its n
    # evaluations, so FRI passes and only the out-of-domain identity fails
    badm = generate_synth_mul_trace(64)
    badm[9, 2] = (int(badm[9, 2]) + 1) % P
    proof = ts.prove(config, SynthMulAir(64), ts.BfChallenger(), badm, [])
    assert orc.verify(orc.FriConfig(2, 6, 8), ts.air_tape(SynthMulAir(64), 0), proof.words, []) == 7


# ------------------------------------------------------------------ BASELINE sizes: properties
@pytest.mark.parametrize("name,make,has_pis,cfg", [
    ("config2_fib_2p20", lambda: (FibonacciAir(), generate_fibonacci_trace(0, 1, 1 << 20)), True, (2, 28, 8)),
    ("config3_mul64_2p20", lambda: (SynthMulAir(64), generate_synth_mul_trace(1 << 20)), False, (2, 28, 8)),
], ids=["config2", "config3"])
def test_full_size_proof_is_accepted(ctx, orc, name, make, has_pis, cfg):
    """At 2^20 rows the oracle prover is too slow to diff against, but its verifier (restated
    from the reference) is cheap: acceptance + determinism + tamper rejection."""
    air, trace = make()
    pis = fibonacci_public_values(trace) if has_pis else np.zeros(0, dtype=np.uint32)
    tape = ts.air_tape(air, len(pis))
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, tape)
    p1 = ts.prove(config, cair, ts.BfChallenger(), trace, pis)
    ocfg = orc.FriConfig(*cfg)
    assert orc.verify(ocfg, tape, p1.words, pis) == 0
    p2 = ts.prove(config, cair, ts.BfChallenger(), trace, pis)
    assert (p1.words == p2.words).all(), "proving is not deterministic"
    bad = p1.words.copy()
    bad[30] = (int(bad[30]) + 1) % P
    assert orc.verify(ocfg, tape, bad, pis) != 0
    assert p1.degree_bits == 20 and len(p1.commit_phase_commits) == 20


# ------------------------------------------------------------------ check_constraints on the GPU
def test_check_constraints_matches_oracle(ctx, orc):
    # reference uni-stark/src/check_constraints.rs:11-39 (what a debug build of prove() runs first)
    fib = FibonacciAir()
    t = generate_fibonacci_trace(0, 1, 1 << 10)
    pis = fibonacci_public_values(t)
    tape = ts.air_tape(fib, 3)
    assert ts.check_constraints(fib, t, pis, ctx) == -1 == orc.check_constraints(tape, t, pis)
    for row, col in ((5, 1), (0, 0), (1023, 1), (700, 0)):
        bad = t.copy()
        bad[row, col] = (int(bad[row, col]) + 1) % P
        assert ts.check_constraints(fib, bad, pis, ctx) == orc.check_constraints(tape, bad, pis) >= 0
    wrong_pis = np.array([0, 1, 5], dtype=np.uint32)
    assert ts.check_constraints(fib, t, wrong_pis, ctx) == orc.check_constraints(tape, t, wrong_pis) >= 0
    for air, tr in ((SynthMulAir(64), generate_synth_mul_trace(1 << 9)),
                    (SynthExtAir(163), generate_synth_ext_trace(1 << 7, 163))):
        tp = ts.air_tape(air, 0)
        assert ts.check_constraints(air, tr, [], ctx) == -1
        bad = tr.copy()
        bad[77, 2] = (int(bad[77, 2]) + 3) % P
        assert ts.check_constraints(air, bad, [], ctx) == orc.check_constraints(tp, bad, []) >= 0


# ------------------------------------------------------------------ the other BASELINE shapes
def test_config4_shape_single_gpu(ctx, orc):
    """BASELINE config 4's shape on ONE GPU (n = 2^22 x 64, log_blowup 4, 16 queries): a 16 GiB LDE,
    the three-round strided NTT plan, 22 FRI rounds; accepted by the oracle's verifier."""
    air = SynthMulAir(64)
    trace = generate_synth_mul_trace(1 << 22)
    tape = ts.air_tape(air, 0)
    cfg = (4, 16, 8)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    proof = ts.prove(config, air, ts.BfChallenger(), trace, [])
    assert orc.verify(orc.FriConfig(*cfg), tape, proof.words, []) == 0
    ts.verify(config, air, ts.BfChallenger(), proof, [])
    assert proof.degree_bits == 22 and len(proof.commit_phase_commits) == 22


def test_config5_shape(ctx, orc):
    """BASELINE config 5's stand-in (SynthExt-163: EF4 column groups, width 163, log_blowup 4,
    16 queries) at 2^18 rows: 652-byte leaves (11 Blake3 blocks) and a wide quotient program."""
    air = SynthExtAir(163)
    trace = generate_synth_ext_trace(1 << 18, 163)
    tape = ts.air_tape(air, 0)
    cfg = (4, 16, 8)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    proof = ts.prove(config, air, ts.BfChallenger(), trace, [])
    assert orc.verify(orc.FriConfig(*cfg), tape, proof.words, []) == 0
    bad = proof.words.copy()
    bad[40] = (int(bad[40]) + 1) % P
    assert orc.verify(orc.FriConfig(*cfg), tape, bad, []) != 0


def test_config5_baseline_shape(ctx, orc):
    """BASELINE config 5 at its full shape: SynthExt-163, 2^20 x 163, log_blowup 4, 16 queries
    (shape from README.md:91,101; the AIR is this build's stand-in).  The trace is
    generated on the device (683 MB never crosses PCIe); the LDE is 10.9 GB.  Accept + determinism +
    tamper rejection, as for configs 2 and 3."""
    air = SynthExtAir(163)
    tape = ts.air_tape(air, 0)
    cfg = (4, 16, 8)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, tape)
    n = 1 << 20
    t = ts.DeviceMatrix.synth_ext(ctx, n, 163)
    assert ts.check_constraints(cair, t, [], ctx) == -1
    p1 = ts.prove(config, cair, ts.BfChallenger(), t, [])
    ocfg = orc.FriConfig(*cfg)
    assert orc.verify(ocfg, tape, p1.words, []) == 0
    ts.verify(config, air, ts.BfChallenger(), p1, [])
    p2 = ts.prove(config, cair, ts.BfChallenger(), ts.DeviceMatrix.synth_ext(ctx, n, 163), [])
    assert (p1.words == p2.words).all(), "proving is not deterministic"
    for pos in (30, len(p1.words) // 2, len(p1.words) - 3):
        bad = p1.words.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert orc.verify(ocfg, tape, bad, []) != 0
    assert p1.degree_bits == 20 and len(p1.commit_phase_commits) == 20
    assert len(p1.query_proofs) == 16 and len(p1.trace_local) == 163


@pytest.mark.parametrize("log_n,w", [(0, 13), (4, 25), (9, 163), (12, 163), (7, 37)])
def test_trace_synth_ext_on_device(ctx, log_n, w):
    n = 1 << log_n
    got = ts.DeviceMatrix.synth_ext(ctx, n, w).download()
    want = generate_synth_ext_trace(n, w)
    assert (got == want).all()


# ------------------------------------------------------------------ general PCS (fri/tests/pcs.rs)
def test_commit_mixed_heights(ctx, orc):
    # one batch, matrices of different heights (bf_mmcs.rs:22-42): shorter matrices are injected
    # into the level with as many nodes as they have rows
    b = 1
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 2, 8), ctx)
    shapes = [(9, 3), (6, 5), (9, 2), (3, 70), (6, 1), (0, 2)]
    mats = [rand_mat(70 + i, 1 << lg, w) for i, (lg, w) in enumerate(shapes)]
    root, data = pcs.commit([((lg, 1), m.copy()) for (lg, _), m in zip(shapes, mats)])
    ldes = [orc.commit_lde(m, 1, b) for m in mats]
    for i in range(len(mats)):
        assert (data.lde(i) == ldes[i]).all()
    om = orc.OracleMmcs(ldes)
    for lvl in range(data.log_height + 1):
        assert (data.digests(lvl) == om.layer(lvl)).all(), f"digest level {lvl} differs"
    assert (root == om.root).all()
    for idx in (0, 1, 513, 1023):
        rows, path = data.open_batch(idx)
        orows, opath = om.open(idx)
        assert (rows == orows).all() and (path == opath).all()
        assert om.verify(idx, rows, path, root)


PCS_SHAPES = (
    [[[i]] for i in range(3, 6)]            # single (pcs.rs:135-142)
    + [[[2, 1]]]                            # small
    + [[[2] * 5]]                           # many_equal
    + [[list(range(3, 3 + i))[::-1]] for i in range(1, 3)]  # many_different_rev
    + [[[3]], [[3], [3]], [[3], [2]], [[2], [3]], [[4, 2], [4, 2]], [[2, 2], [3, 3]],
       [[3, 3], [2, 2]], [[2], [3, 3]]]     # multiple_rounds
    + [[[12, 9, 12], [11, 5]], [[0, 1], [2]], [[14], [10, 12]]]  # beyond the one-workgroup tail
)


@pytest.mark.parametrize("log_blowup", [1, 2])
@pytest.mark.parametrize("shape", PCS_SHAPES, ids=[str(s) for s in PCS_SHAPES])
def test_pcs_open_bit_identical_to_oracle(ctx, orc, log_blowup, shape):
    # fri/tests/pcs.rs:62-90: commit every round, observe, sample zeta, open everything at zeta;
    # roots, opened values and the whole FriProof must equal the oracle's
    cfg = ts.FriConfig(log_blowup, 2, 8)
    ocfg = orc.FriConfig(log_blowup, 2, 8)
    pcs = ts.TwoAdicFriPcs(cfg, ctx)
    seed, evals = 1000, []
    for logs in shape:
        evs = []
        for lg in logs:
            seed += 1
            evs.append(rand_mat(seed, 1 << lg, 2 + seed % 3))
        evals.append(evs)
    oroots, ozeta, oopened, oproof = orc.pcs_commit_open(ocfg, shape, evals)

    ch = ts.BfChallenger()
    datas = []
    for logs, evs in zip(shape, evals):
        root, data = pcs.commit([((lg, 1), e.copy()) for lg, e in zip(logs, evs)])
        datas.append(data)
    for r, data in enumerate(datas):
        assert (data.root == oroots[r]).all(), f"round {r} root differs"
        ch.observe_commitment(data.root)
    zeta = ch.sample()
    assert (zeta == ozeta).all()
    opened, proof = pcs.open([(d, [[zeta]] * d.n_mats) for d in datas], ch)
    flat = np.concatenate([p for r in opened for m in r for p in m])
    assert (flat == oopened).all(), "opened values differ"
    assert len(proof) == len(oproof) and (proof == oproof).all(), "FriProof differs"
    # pcs.rs:92-117: the verifier side, with the product's own Pcs::verify
    vch = ts.BfChallenger()
    for d in datas:
        vch.observe_commitment(d.root)
    assert (vch.sample() == zeta).all()
    claims = [(d.root, [(lg, [(zeta, opened[r][m][0])]) for m, lg in enumerate(logs)])
              for r, (d, logs) in enumerate(zip(datas, shape))]
    pcs.verify(claims, proof, vch)
    assert vch.sample_bits(16) == ch.sample_bits(16)  # prover and verifier transcripts agree


def test_pcs_open_two_points_matches_prove_shape(ctx, orc):
    # the general open on the prove() shape (trace at {zeta, zeta*w}, chunks at {zeta}) must give
    # the fused path's opened values
    log_n, w, b = 8, 6, 2
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 3, 4), ctx)
    tr = rand_mat(5, 1 << log_n, w)
    chunks = [rand_mat(6 + c, 1 << log_n, 4) for c in range(2)]
    g = pow(0x1A427A41, 1 << (27 - (log_n + 1)), P)
    _, td = pcs.commit([((log_n, 1), tr.copy())])
    _, qd = pcs.commit([((log_n, 31 * pow(g, c, P) % P), m.copy()) for c, m in enumerate(chunks)])
    zeta = np.array([5, 6, 7, 8], dtype=np.uint32)
    wn = pow(0x1A427A41, 1 << (27 - log_n), P)
    zeta_next = (zeta.astype(np.uint64) * wn % P).astype(np.uint32)
    ch = ts.BfChallenger()
    alpha = ch.clone().sample()
    opened, proof = pcs.open([(td, [[zeta, zeta_next]]), (qd, [[zeta]] * 2)], ch)
    want, _ = pcs.open_reduce(td, qd, w, zeta, alpha)
    flat = np.concatenate([p for r in opened for m in r for p in m])
    assert (flat == want).all()


# ------------------------------------------------------------------ trace generation on the device
@pytest.mark.parametrize("a,b,log_n", [(0, 1, 0), (0, 1, 3), (3, 5, 7), (P - 1, P - 2, 12), (0, 1, 20)])
def test_trace_fibonacci_on_device(ctx, a, b, log_n):
    # generate_trace_rows (uni-stark/tests/fib_air.rs:59-78), row by row on the host vs in HBM
    n = 1 << log_n
    got = ts.DeviceMatrix.fibonacci(ctx, a, b, n).download()
    want = generate_fibonacci_trace(a, b, n)
    assert got.shape == want.shape and (got == want).all()
    if log_n == 3 and (a, b) == (0, 1):
        assert got[-1, 1] == 21  # fib_air.rs:143


@pytest.mark.parametrize("log_n,w", [(0, 3), (5, 7), (10, 64), (14, 64), (9, 2)])
def test_trace_synth_mul_on_device(ctx, log_n, w):
    n = 1 << log_n
    got = ts.DeviceMatrix.synth_mul(ctx, n, w).download()
    want = generate_synth_mul_trace(n, w)
    assert (got == want).all()


def test_prove_from_device_generated_trace(ctx, orc):
    # the whole path without an H2D of the trace: generate, check constraints, prove; same proof
    # as from the host-generated trace
    n = 1 << 12
    cfg = (2, 9, 8)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    air = SynthMulAir(64)
    dm = ts.DeviceMatrix.synth_mul(ctx, n, 64)
    assert ts.check_constraints(air, dm, [], ctx) == -1
    got = ts.prove(config, air, ts.BfChallenger(), dm, [])
    want = ts.prove(config, air, ts.BfChallenger(), generate_synth_mul_trace(n), [])
    assert (got.words == want.words).all()
    fib = ts.DeviceMatrix.fibonacci(ctx, 0, 1, n)
    host = generate_fibonacci_trace(0, 1, n)
    pis = fibonacci_public_values(host)
    got = ts.prove(config, FibonacciAir(), ts.BfChallenger(), fib, pis)
    want = ts.prove(config, FibonacciAir(), ts.BfChallenger(), host, pis)
    assert (got.words == want.words).all()


# ------------------------------------------------------------------ compiled-language host (C++)
def test_cpp_example_runs_the_reference_test_through_the_c_abi(ctx, orc, tmp_path):
    """examples/fib_air.cpp = uni-stark/tests/fib_air.rs (prove, verify with a fresh challenger) in
    C++ over include/tapstark.h only; its proof is the one the Python binding and the oracle produce."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_abi_cpu import _build_example
    exe = _build_example(tmp_path)
    for log_n in (3, 10):
        out_bin = str(tmp_path / f"proof{log_n}.bin")
        r = subprocess.run([exe, str(log_n), out_bin], capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "verify -> 0, with a wrong public value -> 7" in r.stdout
        words = np.fromfile(out_bin, dtype=np.uint32)
        trace = generate_fibonacci_trace(0, 1, 1 << log_n)
        pis = fibonacci_public_values(trace)
        if log_n == 3:
            assert "public values [0, 1, 21]" in r.stdout  # fib_air.rs:143
        want = orc.prove(orc.FriConfig(2, 28, 8), ts.air_tape(FibonacciAir(), 3), trace, pis)
        assert len(words) == len(want) and (words == want).all()


def test_cpp_stream_example_lanes_gate_and_pinned_uploads(ctx, orc, tmp_path):
    """examples/prove_stream.cpp: S lanes (one thread + one context each), the start gate, traces born
    on the device or uploaded asynchronously from page-locked host memory -- what the (unbuilt) Rust
    prove_gpu_stream does, from a compiled language over include/tapstark.h only.  Every proof of the
    run equals the oracle's proof of that trace."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_abi_cpu import _build_example
    exe = _build_example(tmp_path, "prove_stream")
    log_n = 11
    trace = generate_synth_mul_trace(1 << log_n)
    want = orc.prove(orc.FriConfig(2, 28, 8), ts.air_tape(SynthMulAir(64), 0), trace, [])
    for mode in ("device", "pinned"):
        out_bin = str(tmp_path / f"stream_{mode}.bin")
        r = subprocess.run([exe, str(log_n), "12", "3", mode, out_bin], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "all proofs identical" in r.stdout and "verify -> 0" in r.stdout
        words = np.fromfile(out_bin, dtype=np.uint32)
        assert len(words) == len(want) and (words == want).all(), mode


def test_cpp_sharded_example_eight_ranks_through_the_c_abi(ctx, orc, tmp_path):
    """examples/prove_sharded.cpp: ONE proof over G thread-ranks from a compiled language over
    include/tapstark.h only (what the unbuilt Rust prove_gpu_sharded does): config 4's split (log_blowup
    4, two cosets per rank at G = 8), the in-process communicator, quotient chunks broadcast or computed
    locally.  The program itself checks every rank against ts_prove; here its proof is also the oracle's."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_abi_cpu import _build_example
    exe = _build_example(tmp_path, "prove_sharded")
    log_n = 11
    trace = generate_synth_mul_trace(1 << log_n)
    want = orc.prove(orc.FriConfig(4, 16, 8), ts.air_tape(SynthMulAir(64), 0), trace, [])
    for G, mode in ((8, "bcast"), (8, "localq"), (2, "localq")):
        out_bin = str(tmp_path / f"sharded_{G}_{mode}.bin")
        r = subprocess.run([exe, str(log_n), str(G), "local", mode, out_bin], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "every rank's proof equals ts_prove's" in r.stdout and "verify -> 0" in r.stdout
        words = np.fromfile(out_bin, dtype=np.uint32)
        assert len(words) == len(want) and (words == want).all(), (G, mode)


def test_fri_fold_device_vectors_match_oracle(ctx, orc):
    # ts_fri_fold_device: fold_matrix (two_adic_pcs.rs:116-147) on vectors that already live in HBM,
    # the entry point `bench.py --workload fold` times (fri/benches/fold_even_odd.rs sizes)
    import ctypes as C

    import torch

    from tapstark_amd import _lib
    l = _lib.lib()
    rng = np.random.default_rng(5)
    for log_size in (1, 2, 5, 12, 17):
        n = 1 << log_size
        vec = rng.integers(0, P, (n, 4), dtype=np.uint32)
        beta = rng.integers(0, P, 4, dtype=np.uint32)
        d_in = torch.from_numpy(vec.view(np.int32)).to("cuda:0")
        d_out = torch.zeros((n // 2, 4), dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        ctx.check(l.ts_fri_fold_device(ctx.h, d_in.data_ptr(), n // 2, beta.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       d_out.data_ptr()))
        ctx.synchronize()
        assert (d_out.cpu().numpy().view(np.uint32) == orc.fold_matrix(vec, beta)).all(), log_size
    # misaligned device pointers are refused (EF4 = 16-byte accesses)
    rc = l.ts_fri_fold_device(ctx.h, d_in.data_ptr() + 4, 1, beta.ctypes.data_as(C.POINTER(C.c_uint32)), d_out.data_ptr())
    assert rc == 1


# ------------------------------------------------------------------ FRI alone (fri/tests/fri.rs)
@pytest.mark.parametrize("perm,ext,cfg,degs", [
    (1, False, (1, 10, 8), range(1, 10)),        # test_compelte_fri_process, fri.rs:51-147
    (0, True, (1, 10, 8), range(1, 10)),
    (0, True, (2, 7, 8), [3, 9, 14, 15]),        # beyond the one-workgroup tail; gaps between heights
])
def test_fri_prove_alone_matches_oracle(ctx, orc, perm, ext, cfg, degs):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_abi_cpu import _fri_rs_inputs
    ins = _fri_rs_inputs(orc, cfg[0], degs)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx)
    pch = ts.BfChallenger(perm, ext)
    proof = pcs.fri_prove(ins, pch)
    och = orc.OracleChallenger(perm_kind=perm, sample_ext=ext)
    want = orc.fri_prove(orc.FriConfig(*cfg), ins, och)
    assert len(proof) == len(want) and (proof == want).all()
    vch = ts.BfChallenger(perm, ext)
    pcs.fri_verify(proof, vch)
    assert pch.sample_bits(8) == vch.sample_bits(8) == och.sample_bits(8)  # fri.rs:141-146


MULTI_SHAPES = [[[5, 5, 5]], [[6, 3], [4, 4, 2]], [[3], [7, 7], [5]], [[11, 8, 11, 2]]]


@pytest.mark.parametrize("shape", MULTI_SHAPES, ids=[str(s) for s in MULTI_SHAPES])
def test_pcs_open_several_points_per_matrix(ctx, orc, shape):
    # two_adic_pcs.rs:344-387 with 1, 2 or 3 points per matrix (matrix k: zeta * 7^j, j < 1 + k % 3),
    # mixed heights and several rounds: opened values, FriProof and the verifier
    cfg = (1, 4, 8)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx)
    seed, evals = 3000, []
    for logs in shape:
        evs = []
        for lg in logs:
            seed += 1
            evs.append(rand_mat(seed, 1 << lg, 1 + seed % 5))
        evals.append(evs)
    oroots, ozeta, oopened, oproof = orc.pcs_commit_open(orc.FriConfig(*cfg), shape, evals, multi=True)
    ch = ts.BfChallenger()
    datas = [pcs.commit([((lg, 1), e.copy()) for lg, e in zip(logs, evs)])[1]
             for logs, evs in zip(shape, evals)]
    for d in datas:
        ch.observe_commitment(d.root)
    zeta = ch.sample()
    assert (zeta == ozeta).all()

    def points(k):
        return [(zeta.astype(np.uint64) * pow(7, j, P) % P).astype(np.uint32) for j in range(1 + k % 3)]

    rounds, k = [], 0
    for d in datas:
        pts = []
        for _ in range(d.n_mats):
            pts.append(points(k))
            k += 1
        rounds.append((d, pts))
    vch = ch.clone()
    opened, proof = pcs.open(rounds, ch)
    flat = np.concatenate([p for r in opened for m in r for p in m])
    assert (flat == oopened).all(), "opened values differ"
    assert len(proof) == len(oproof) and (proof == oproof).all(), "FriProof differs"
    claims = [(d.root, [(lg, list(zip(pts[m], opened[r][m]))) for m, lg in enumerate(logs)])
              for r, ((d, pts), logs) in enumerate(zip(rounds, shape))]
    pcs.verify(claims, proof, vch)


# ------------------------------------------------------------------ edges of the shape space
EDGE_CASES = [
    ("mul64-n1", lambda: (SynthMulAir(64), generate_synth_mul_trace(1)), False, (2, 5, 8)),
    ("fib-n1", lambda: (FibonacciAir(), generate_fibonacci_trace(0, 1, 1)), True, (2, 4, 8)),
    ("mul3-n4-b1-nopow", lambda: (SynthMulAir(3), generate_synth_mul_trace(4, 3)), False, (1, 3, 0)),
    ("mul64-q100", lambda: (SynthMulAir(64), generate_synth_mul_trace(1 << 10)), False, (2, 100, 8)),
    ("mul64-b5", lambda: (SynthMulAir(64), generate_synth_mul_trace(1 << 8)), False, (5, 4, 8)),
    ("mul200", lambda: (SynthMulAir(200), generate_synth_mul_trace(1 << 6, 200)), False, (2, 4, 8)),
    # wider than the mailbox page's quarter (8w + 16qd > 4096 words): the opened-value sums take the
    # large-message path (csrc/context.cpp Context::mailbox)
    ("mul640", lambda: (SynthMulAir(640), generate_synth_mul_trace(1 << 5, 640)), False, (2, 4, 8)),
    ("mul1500", lambda: (SynthMulAir(1500), generate_synth_mul_trace(1 << 3, 1500)), False, (2, 3, 4)),
]


@pytest.mark.parametrize("name,make,has_pis,cfg", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_shapes_bit_identical(ctx, orc, name, make, has_pis, cfg):
    # one-row traces, no proof of work, 100 queries, blowup 32, a 200-column row (13 Blake3 blocks)
    air, trace = make()
    pis = fibonacci_public_values(trace) if has_pis else np.zeros(0, dtype=np.uint32)
    tape = ts.air_tape(air, len(pis))
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    proof = ts.prove(config, air, ts.BfChallenger(), trace, pis)
    want = orc.prove(orc.FriConfig(*cfg), tape, trace, pis)
    assert len(proof.words) == len(want) and (proof.words == want).all()
    ts.verify(config, air, ts.BfChallenger(), proof, pis)


@pytest.mark.parametrize("bits", [0, 1, 5, 8, 10, 11])
def test_proof_of_work_witness_found_on_the_device(ctx, orc, bits):
    # fri/src/prover.rs:43 `challenger.grind`: the last commit-phase kernel tries 256 candidates a pass and
    # hands the host the smallest witness that passes (csrc/fri.hip: k_fri_tail); the host confirms it with
    # one sponge step.  Same witness, same proof as the oracle's serial search -- over several transcripts,
    # for bit counts whose witnesses lie in the first pass (<= 8 bits, mostly) and beyond it (10, 11 bits:
    # one candidate in ~340 / ~680 passes)
    seen = []
    stat0 = [ctx.stat(k) for k in (6, 7, 8)]
    n_proved = 0
    for a0 in range(4):
        trace = generate_fibonacci_trace(a0, 1, 1 << 6)
        pis = fibonacci_public_values(trace)
        tape = ts.air_tape(FibonacciAir(), len(pis))
        cfg = (2, 3, bits)
        want = None
        try:
            want = orc.prove(orc.FriConfig(*cfg), tape, trace, pis)
        except Exception:  # no witness below 4096: the product must refuse too
            with pytest.raises(ts._lib.TsError):
                ts.prove(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx)), FibonacciAir(), ts.BfChallenger(), trace, pis)
            continue
        proof = ts.prove(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx)), FibonacciAir(), ts.BfChallenger(), trace, pis)
        assert len(proof.words) == len(want) and (proof.words == want).all()
        seen.append(proof.pow_witness)
        n_proved += 1
    assert seen, "no transcript had a witness"
    # the witness really came from k_fri_tail's search: every proof's candidate was accepted by the host's
    # one-step check, none refused, and the host never searched (ts_ctx_stat 6 / 7 / 8)
    accepted, rejected, host = (ctx.stat(k) - s0 for k, s0 in zip((6, 7, 8), stat0))
    assert rejected == 0
    if os.environ.get("TS_HOST_GRIND"):
        assert accepted == 0 and host == n_proved
    else:
        assert accepted == n_proved and host == 0, (accepted, host, n_proved)
    if bits == 0:
        assert seen == [0] * len(seen)
    if bits >= 10:
        assert max(seen) >= 256, f"no witness beyond the first pass was exercised: {seen}"


def test_zero_queries_refused(ctx):
    with pytest.raises(ts._lib.TsError):
        ts.prove(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 0, 8), ctx)), SynthMulAir(64),
                 ts.BfChallenger(), generate_synth_mul_trace(4), [])


def test_device_out_of_memory_is_a_status_not_a_crash():
    # more than the card holds (2^26 x 2048 words = 512 GiB against 288 GB of HBM): the allocation is
    # refused with TS_ERR_OOM (after the context drops its block cache and retries once), nothing
    # has been launched on the missing buffer, and the context goes on working
    from tapstark_amd._lib import TsError

    c = ts.Context(0)
    with pytest.raises(TsError) as e:
        ts.DeviceMatrix.synth_mul(c, 1 << 26, 2048)
    assert e.value.code == 3  # TS_ERR_OOM
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 8), c))
    trace = generate_fibonacci_trace(0, 1, 64)
    pis = fibonacci_public_values(trace)
    proof = ts.prove(config, FibonacciAir(), ts.BfChallenger(), trace, pis)
    ts.verify(config, FibonacciAir(), ts.BfChallenger(), proof, pis)


def test_out_of_memory_in_the_middle_of_a_proof():
    # a 64 GiB trace fits, its 4x LDE does not: prove() must unwind (buffers back to the pool, kernels
    # already queued on them finish harmlessly) and report TS_ERR_OOM; the context stays usable
    from tapstark_amd._lib import TsError

    c = ts.Context(0)
    m = ts.DeviceMatrix.synth_mul(c, 1 << 24, 1024)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 4, 8), c))
    with pytest.raises(TsError) as e:
        ts.prove(config, ts.CompiledAir(c, ts.air_tape(SynthMulAir(1024), 0)), ts.BfChallenger(), m, [])
    assert e.value.code == 3  # TS_ERR_OOM
    trace = generate_fibonacci_trace(0, 1, 64)
    pis = fibonacci_public_values(trace)
    proof = ts.prove(config, FibonacciAir(), ts.BfChallenger(), trace, pis)
    ts.verify(config, FibonacciAir(), ts.BfChallenger(), proof, pis)
    del proof, m, c  # the context (and its 128 GB block cache) goes away here


# ------------------------------------------------------------------ several proofs in flight, one call
def test_prove_stream_is_prove_on_several_lanes(ctx, orc):
    """ts_prove_stream: independent proofs on several contexts, the lane threads inside the library.  Every
    proof is prove()'s (checked as the returned last proof over runs that end on each lane, and through the
    start / wall time arrays that every proof ran), a consumed trace is an error of that call, and the gate
    spaces the starts."""
    air = SynthMulAir(7)
    tape = ts.air_tape(air, 0)
    cfg = (2, 5, 4)
    ctxs = [ctx, ts.Context(0), ts.Context(0)]
    lanes = [(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c)), ts.CompiledAir(c, tape)) for c in ctxs]
    ocfg = orc.FriConfig(*cfg)

    def trace(i):
        return generate_synth_mul_trace(1 << 7, 7, seed=1000 + i)

    for n in (1, 3, 7, 8):
        lane_of = [(i * 2 + 1) % 3 for i in range(n)]
        mats = [ts.DeviceMatrix.upload(ctxs[lane_of[i]], trace(i)) for i in range(n)]
        proof, start, wall = ts.prove_stream(lanes, mats, lane_of, [], gate_ms=0.3)
        want = orc.prove(ocfg, tape, trace(n - 1), [])
        assert len(proof.words) == len(want) and (proof.words == want).all(), f"n = {n}"
        assert len(start) == n and (wall > 0).all()
        s = np.sort(start)
        # (the start stamp is taken just after the gate: allow for a thread being descheduled in between)
        assert (np.diff(s) >= 0.2).all(), "two proofs started within the gate"
        with pytest.raises(ts._lib.TsError):  # the matrices are spent
            ts.prove_stream(lanes, mats[:1], lane_of[:1], [])
    # nothing to do is not an error
    proof, start, wall = ts.prove_stream(lanes, [], [], [])
    assert len(proof.words) == 0 and len(start) == 0
