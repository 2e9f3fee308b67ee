"""CPU tests of the oracle itself: reference known answers, independent re-derivations of each
numeric stage, and the reference's own test shapes (prove->verify, FRI transcript equality,
fold_even_odd property, PCS shape matrix).  No GPU."""
import json
import os

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd.airs import (FibonacciAir, SynthExtAir, SynthMulAir, fibonacci_public_values,
                               generate_fibonacci_trace, generate_synth_ext_trace,
                               generate_synth_mul_trace, splitmix64_stream)

P = 0x78000001
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rand_mat(seed, h, w):
    return splitmix64_stream(seed, h * w).reshape(h, w)


# ----------------------------------------------------------- reference KATs
def test_blake3_reference_kats(orc):
    kats = json.load(open(os.path.join(GOLDEN, "kats.json")))
    for k in kats["blake3"]:
        data = np.asarray(k["input_words"], dtype=np.uint32).tobytes()
        assert orc.blake3(data).hex() == k["digest"], k["source"]


def _pattern(n):
    return bytes(i % 251 for i in range(n))


def test_blake3_official_vectors(orc):
    # the official BLAKE3 test-vector set (input byte i % 251, lengths straddling every block and
    # chunk boundary up to 100 chunks), digests from the BLAKE3 team's C code as vendored in LLVM
    # (tests/golden/make_blake3_vectors.py): pins multi-block and multi-chunk hashing -- what every
    # Merkle leaf uses -- on something that is not this repository's own output
    vec = json.load(open(os.path.join(GOLDEN, "blake3_official.json")))
    assert len(vec["official_lengths"]) == 35
    for group in ("official_lengths", "word_lengths"):
        for ln, hexd in vec[group].items():
            assert orc.blake3(_pattern(int(ln))).hex() == hexd, ln


def test_blake3_multiblock_multichunk_self_consistency(orc):
    # lengths straddling block (64) and chunk (1024) boundaries must all differ and be stable
    seen = set()
    for ln in [0, 1, 63, 64, 65, 127, 128, 1023, 1024, 1025, 2048, 2049, 3072, 5000]:
        d = orc.blake3(bytes((i * 7 + 3) & 0xFF for i in range(ln)))
        assert d not in seen
        seen.add(d)
    golden = json.load(open(os.path.join(GOLDEN, "oracle_fixtures.json")))
    for ln, hexd in golden["blake3_pattern"].items():
        assert orc.blake3(bytes((i * 7 + 3) & 0xFF for i in range(int(ln)))).hex() == hexd


def test_challenger_reference_kat(orc):
    kats = json.load(open(os.path.join(GOLDEN, "kats.json")))["challenger_base"]
    word = int.from_bytes(bytes(kats["observe_bytes"]), "little")
    c = orc.OracleChallenger(0, sample_ext=False)
    c.observe(word)
    c.sample_base()
    c.observe(word)
    assert c.sample_base() == kats["second_sample"]


def test_challenger_ext_flow(orc):
    # flow of reference script_expr/src/challenger_expr.rs:322-331; values from SURVEY App. B
    word = int.from_bytes(bytes([1, 2, 3, 4]), "little")
    c = orc.OracleChallenger(0, sample_ext=True)
    c.observe(word)
    assert c.sample().tolist() == [1883249845, 479209046, 520648298, 1319656751]
    c.observe(word)
    assert c.sample().tolist() == [1113150497, 241879703, 420917749, 1543862539]
    c.observe(word)
    assert c.sample_bits(31) == 973543621
    assert c.n_perms == 3


def test_constants(orc):
    k = json.load(open(os.path.join(GOLDEN, "kats.json")))["constants"]
    assert k["p"] == P
    assert (31 * k["generator_inverse"]) % P == 1
    assert pow(31, 15, P) == 0x1A427A41
    assert pow(0x1A427A41, 1 << 26, P) == P - 1  # order exactly 2^27
    t = generate_fibonacci_trace(0, 1, 8)
    assert int(t[-1, 1]) == k["fib_2pow3_last_right"]


def test_grind_smallest_witness(orc):
    c = orc.OracleChallenger()
    c.observe_digest(np.arange(8, dtype=np.uint32))
    c2 = orc.OracleChallenger()
    c2.observe_digest(np.arange(8, dtype=np.uint32))
    w = c.grind(8)
    for cand in range(w):
        cc = orc.OracleChallenger()
        cc.observe_digest(np.arange(8, dtype=np.uint32))
        assert not cc.check_witness(8, cand)
    assert c2.check_witness(8, w)
    assert (c.state_words() == c2.state_words()).all()


# -------------------------------------------------------------- DFT and LDE
@pytest.mark.parametrize("log_n,w", [(0, 1), (1, 2), (3, 2), (6, 5)])
def test_fast_dft_matches_definition(orc, log_n, w):
    m = rand_mat(1, 1 << log_n, w)
    assert (orc.dft_batch(m) == orc.naive_dft(m)).all()
    assert (orc.dft_batch(m, inverse=True) == orc.naive_dft(m, inverse=True)).all()
    assert (orc.dft_batch(orc.dft_batch(m), inverse=True) == m).all()


def test_commit_lde_is_evaluation_of_interpolant(orc):
    # committed row bitrev(j) holds p(31 * omega_N^j)  (SURVEY section 8 row a3), for the trace
    # domain (shift 1) and for a quotient-chunk domain (shift 31*omega)
    log_n, b = 3, 2
    n, N = 1 << log_n, 1 << (log_n + b)
    m = rand_mat(2, n, 2)
    gN = pow(0x1A427A41, 1 << (27 - (log_n + b)), P)
    for dom_shift in (1, 31 * pow(0x1A427A41, 1 << (27 - (log_n + 1)), P) % P):
        lde = orc.commit_lde(m, dom_shift, b)
        for j in range(N):
            x = 31 * pow(gN, j, P) % P
            r = int(f"{j:0{log_n + b}b}"[::-1], 2)
            for c in range(2):
                assert lde[r, c] == orc.eval_interpolant_naive(m[:, c], dom_shift, x)


def test_low_coset_rows_are_the_quotient_domain(orc):
    # first n*qd bit-reversed rows = evaluations on 31*H_{n*qd} (two_adic_pcs.rs:247-258)
    log_n, b = 3, 2
    m = rand_mat(3, 1 << log_n, 1)
    lde = orc.commit_lde(m, 1, b)
    for lq in (0, 1, 2):
        qn = 1 << (log_n + lq)
        g = pow(0x1A427A41, 1 << (27 - (log_n + lq)), P)
        for i in range(qn):
            r = int(f"{i:0{log_n + lq}b}"[::-1], 2) if log_n + lq else 0
            assert lde[r, 0] == orc.eval_interpolant_naive(m[:, 0], 1, 31 * pow(g, i, P) % P)


def test_fold_even_odd_property(orc):
    # reference fri/src/fold_even_odd.rs:64-95: fold(bitrev(DFT(c)), beta) == bitrev(DFT(c_even)
    # + beta*DFT(c_odd)), here with an extension-field beta and EF4-embedded base coefficients
    log_n = 6
    n = 1 << log_n
    coeffs = rand_mat(4, n, 1)
    evals = orc.dft_batch(coeffs)[:, 0]
    even = orc.dft_batch(coeffs[0::2])[:, 0].astype(np.uint64)
    odd = orc.dft_batch(coeffs[1::2])[:, 0].astype(np.uint64)
    beta = np.array([5, 0, 0, 0], dtype=np.uint32)
    expected = (even + 5 * odd) % P
    br = lambda i, bits: int(f"{i:0{bits}b}"[::-1], 2)
    vec = np.zeros((n, 4), dtype=np.uint32)
    for i in range(n):
        vec[i, 0] = evals[br(i, log_n)]
    folded = orc.fold_matrix(vec, beta)
    for i in range(n // 2):
        assert folded[i, 0] == expected[br(i, log_n - 1)]
        assert not folded[i, 1:].any()


def test_fold_row_agrees_with_fold_matrix(orc):
    h = 16
    vec = rand_mat(5, 2 * h, 4)
    beta = rand_mat(6, 1, 4)[0]
    folded = orc.fold_matrix(vec, beta)
    for i in range(h):
        assert (orc.fold_row(i, 4, beta, vec[2 * i], vec[2 * i + 1]) == folded[i]).all()


# --------------------------------------------------------------------- MMCS
def test_mmcs_commit_open_verify_equal_heights(orc):
    mats = [rand_mat(7, 16, 3), rand_mat(8, 16, 4)]
    m = orc.OracleMmcs(mats)
    # leaf = Blake3(row0 || row1) ; node = Blake3(l || r)
    leaf0 = orc.blake3(np.concatenate([mats[0][0], mats[1][0]]).tobytes())
    assert m.layer(0)[0].tobytes() == leaf0
    n0 = orc.blake3(m.layer(0)[0].tobytes() + m.layer(0)[1].tobytes())
    assert m.layer(1)[0].tobytes() == n0
    for idx in (0, 5, 15):
        rows, path = m.open(idx)
        assert (rows == np.concatenate([mats[0][idx], mats[1][idx]])).all()
        assert m.verify(idx, rows, path)
        bad = rows.copy()
        bad[0] ^= 1
        assert not m.verify(idx, bad, path)
        assert not m.verify(idx ^ 1, rows, path)


def test_mmcs_mixed_heights(orc):
    # bf_mmcs.rs:10-15: shorter matrices are opened at index >> (log_max - log_h)
    mats = [rand_mat(9, 8, 2), rand_mat(10, 2, 3), rand_mat(11, 8, 1), rand_mat(12, 4, 2)]
    m = orc.OracleMmcs(mats)
    for idx in range(8):
        rows, path = m.open(idx)
        exp = np.concatenate([mats[0][idx], mats[1][idx >> 2], mats[2][idx], mats[3][idx >> 1]])
        assert (rows == exp).all()
        assert m.verify(idx, rows, path)
        bad = rows.copy()
        bad[2] ^= 1  # inside the height-2 matrix
        assert not m.verify(idx, bad, path)


# ----------------------------------------------------------------------- AIR
def test_symbolic_degrees_match_reference_rules(orc):
    fib = FibonacciAir()
    tape = ts.air_tape(fib, 3)
    assert orc.tape_validate(tape) == 0
    # is_first(1) * (main(1) - public(0)) = 2 ; is_transition has degree 0
    assert ts.get_max_constraint_degree(fib, 3) == 2 == orc.max_constraint_degree(tape)
    assert ts.get_log_quotient_degree(fib, 3) == 0 == orc.log_quotient_degree(tape)
    mul = SynthMulAir(64)
    mt = ts.air_tape(mul, 0)
    assert ts.get_max_constraint_degree(mul, 0) == 3 == orc.max_constraint_degree(mt)
    assert ts.get_log_quotient_degree(mul, 0) == 1 == orc.log_quotient_degree(mt)
    ext = SynthExtAir(163)
    et = ts.air_tape(ext, 0)
    assert orc.max_constraint_degree(et) == 2 and orc.log_quotient_degree(et) == 0


def test_check_constraints(orc):
    fib = FibonacciAir()
    tape = ts.air_tape(fib, 3)
    t = generate_fibonacci_trace(0, 1, 16)
    pis = fibonacci_public_values(t)
    assert orc.check_constraints(tape, t, pis) == -1
    bad = t.copy()
    bad[5, 1] += 1
    assert orc.check_constraints(tape, bad, pis) >= 0
    assert orc.check_constraints(tape, t, np.array([0, 1, 123], dtype=np.uint32)) >= 0
    for air, gen in ((SynthMulAir(64), generate_synth_mul_trace),
                     (SynthMulAir(7), lambda n: generate_synth_mul_trace(n, 7)),
                     (SynthExtAir(163), generate_synth_ext_trace),
                     (SynthExtAir(25), lambda n: generate_synth_ext_trace(n, 25))):
        tr = gen(32)
        assert orc.check_constraints(ts.air_tape(air, 0), tr, []) == -1


# ------------------------------------------------------ prove -> verify round trips
def test_fib_air_prove_verify_reference_shape(orc):
    # reference uni-stark/tests/fib_air.rs:117-149: n = 2^3, log_blowup 2, 28 queries, 8 PoW bits
    air = FibonacciAir()
    trace = generate_fibonacci_trace(0, 1, 1 << 3)
    pis = fibonacci_public_values(trace)
    assert pis.tolist() == [0, 1, 21]
    tape = ts.air_tape(air, 3)
    for q in (28, 16, 6):  # BASELINE config 1 says 16; fib_air.rs:153 uses 6
        cfg = orc.FriConfig(2, q, 8)
        proof = orc.prove(cfg, tape, trace, pis)
        assert orc.verify(cfg, tape, proof, pis) == 0
    golden = json.load(open(os.path.join(GOLDEN, "oracle_fixtures.json")))
    cfg = orc.FriConfig(2, 28, 8)
    proof = orc.prove(cfg, tape, trace, pis)
    assert orc.blake3(proof.tobytes()).hex() == golden["fib8_q28_proof_blake3"]
    tr = orc.last_transcript()
    assert tr["alpha"].tolist() == golden["fib8_q28_alpha"]
    assert tr["zeta"].tolist() == golden["fib8_q28_zeta"]


@pytest.mark.parametrize("log_n", [4, 7])
def test_prove_verify_wider_airs(orc, log_n):
    n = 1 << log_n
    cfg = orc.FriConfig(2, 10, 8)
    for air, tr in ((SynthMulAir(64), generate_synth_mul_trace(n)),
                    (SynthMulAir(7), generate_synth_mul_trace(n, 7)),
                    (SynthExtAir(25), generate_synth_ext_trace(n, 25))):
        tape = ts.air_tape(air, 0)
        proof = orc.prove(cfg, tape, tr, [])
        assert orc.verify(cfg, tape, proof, []) == 0


def test_verify_rejects_tampering(orc):
    air = FibonacciAir()
    trace = generate_fibonacci_trace(0, 1, 16)
    pis = fibonacci_public_values(trace)
    tape = ts.air_tape(air, 3)
    cfg = orc.FriConfig(2, 8, 8)
    proof = orc.prove(cfg, tape, trace, pis)
    assert orc.verify(cfg, tape, proof, pis) == 0
    # header(5) | trace root(8) | quotient root(8) | trace_local(8) trace_next(8) chunks(16)
    for pos, codes in ((5, (7, 8, 4, 3)), (13, (8, 4, 3, 7)), (21, (7, 8)), (21 + 8 + 8, (7, 8)),
                       (len(proof) - 1, (3,)), (len(proof) - 5, (6,))):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        rc = orc.verify(cfg, tape, bad, pis)
        assert rc != 0, pos
    assert orc.verify(cfg, tape, proof, np.array([0, 1, 5], dtype=np.uint32)) == 7
    assert orc.verify(orc.FriConfig(2, 9, 8), tape, proof, pis) == 2
    assert orc.verify(cfg, tape, proof[:-3], pis) == 9
    # an invalid trace is refused before proving (prover.rs:40-41 debug check)
    bad_trace = trace.copy()
    bad_trace[3, 0] += 1
    with pytest.raises(RuntimeError):
        orc.prove(cfg, tape, bad_trace, pis)


# ------------------------------------------------- reference fri/tests/fri.rs shape
@pytest.mark.parametrize("sample_ext", [False, True])
def test_complete_fri_process(orc, sample_ext):
    # fri.rs:52-147: polys of 2^1..2^9, blowup 2 (log 1), 10 queries, TestPermutation,
    # pass-through input proof, prover/verifier transcripts end equal
    cfg = orc.FriConfig(1, 10, 8)
    inputs = []
    for deg_bits in range(9, 0, -1):
        ev = rand_mat(100 + deg_bits, 1 << deg_bits, 4 if sample_ext else 1)
        lde = orc.commit_lde(ev, 1, 1)  # coset_lde_batch(evals, 1, generator) + bit reversal
        vec = np.zeros((lde.shape[0], 4), dtype=np.uint32)
        vec[:, : lde.shape[1]] = lde
        if sample_ext:
            # an EF-valued codeword: each of the 4 coefficient columns is a low-degree poly
            pass
        inputs.append(vec)
    assert orc.fri_roundtrip(cfg, inputs, sample_ext=sample_ext, perm_kind=1) == 0
    assert orc.fri_roundtrip(cfg, inputs, sample_ext=sample_ext, perm_kind=0) == 0
    # a non-low-degree input must trip the final-poly assertion (fri/src/prover.rs:130-134)
    bad = [v.copy() for v in inputs]
    bad[0][3, 0] ^= 1
    assert orc.fri_roundtrip(cfg, bad, sample_ext=sample_ext, perm_kind=1) == -5


# ------------------------------------------------- reference fri/tests/pcs.rs shapes
PCS_SHAPES = (
    [[[i]] for i in range(3, 6)]            # single
    + [[[2, 1]]]                            # small
    + [[[2] * 5]]                           # many_equal
    + [[list(range(3, 3 + i))[::-1]] for i in range(1, 3)]  # many_different_rev
    + [[[3]], [[3], [3]], [[3], [2]], [[2], [3]], [[4, 2], [4, 2]], [[2, 2], [3, 3]],
       [[3, 3], [2, 2]], [[2], [3, 3]]]     # multiple_rounds
)


@pytest.mark.parametrize("log_blowup", [1, 2])
@pytest.mark.parametrize("shape", PCS_SHAPES, ids=[str(s) for s in PCS_SHAPES])
def test_pcs_shapes(orc, log_blowup, shape):
    # pcs.rs:135-181 with num_queries 2, 8 PoW bits (:204-212); widths 2..4 (:52)
    cfg = orc.FriConfig(log_blowup, 2, 8)
    seed = 1000
    evals = []
    for logs in shape:
        evs = []
        for lg in logs:
            seed += 1
            evs.append(rand_mat(seed, 1 << lg, 2 + seed % 3))
        evals.append(evs)
    assert orc.pcs_roundtrip(cfg, shape, evals) == 0
    assert orc.pcs_roundtrip(cfg, shape, evals, tamper=1) != 0
    assert orc.pcs_roundtrip(cfg, shape, evals, tamper=2) != 0
