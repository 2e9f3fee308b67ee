"""CPU-side checks of the product library: it loads, exports every symbol include/tapstark.h
declares, and its host-only parts (Fiat-Shamir challenger, AIR tape compiler) agree with the
oracle.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd import _lib
from tapstark_amd.airs import FibonacciAir, SynthMulAir

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from tapstark_amd.build import build

    build()
    return _lib.lib()


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "tapstark.h")).read()
    declared = set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", hdr)) - {"ts_status"}
    assert declared == set(_lib.ABI_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ts_abi_version() == 5


def test_device_count_and_null_context_diagnostics(lib):
    import ctypes as C

    import torch

    n = lib.ts_device_count()
    assert n == (torch.cuda.device_count() if torch.cuda.is_available() else 0)
    # the diagnostics refuse a missing context with a status, like every entry point
    v, ms = C.c_uint64(), C.c_double()
    assert lib.ts_ctx_stat(None, 0, C.byref(v)) == 1            # TS_ERR_INVALID
    assert lib.ts_bench_stage(None, 0, 10, 4, 1, 1, C.byref(ms)) == 1


def test_no_cpu_fallback_without_device(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.TsError):
        ts.Context(0)


def test_challenger_matches_oracle(lib, orc):
    rng = np.random.default_rng(1)
    for perm in (0, 1):
        for ext in (True, False):
            a = ts.BfChallenger(perm, ext)
            b = orc.OracleChallenger(perm, ext)
            for step in range(200):
                k = rng.integers(0, 4)
                if k == 0:
                    w = int(rng.integers(0, 2**32))
                    a.observe(w)
                    b.observe(w)
                elif k == 1:
                    d = rng.integers(0, 2**32, size=8, dtype=np.uint64).astype(np.uint32)
                    a.observe_commitment(d)
                    b.observe_digest(d)
                elif k == 2:
                    assert a.sample().tolist() == b.sample().tolist()
                else:
                    bits = int(rng.integers(1, 28))
                    assert a.sample_bits(bits) == b.sample_bits(bits)
            if perm == 0:
                assert a.grind(8) == b.grind(8)
            st = a.state()
            assert st[:16].tolist() == list(b.c.state)
            assert int(st[16]) == b.c.n_in and int(st[25]) == b.c.n_out


def test_challenger_reference_kat(lib):
    # reference script_expr/src/challenger_expr.rs:279-296
    word = int.from_bytes(bytes([1, 1, 1, 1]), "little")
    c = ts.BfChallenger(0, sample_ext=False)
    c.observe(word)
    c.sample()
    c.observe(word)
    assert int(c.sample()[0]) == 1103171332


def test_clone_is_independent(lib):
    a = ts.BfChallenger()
    a.observe(7)
    b = a.clone()
    assert a.sample().tolist() == b.sample().tolist()
    a.observe(1)
    assert a.state().tolist() != b.state().tolist()


# ------------------------------------------------------------------ native verifier (host only)
def _cases():
    from tapstark_amd.airs import (SynthExtAir, fibonacci_public_values, generate_fibonacci_trace,
                                   generate_synth_ext_trace, generate_synth_mul_trace)
    t = generate_fibonacci_trace(0, 1, 8)
    yield "fib8", FibonacciAir(), t, fibonacci_public_values(t), (2, 28, 8)
    t = generate_fibonacci_trace(2, 7, 64)
    yield "fib64_b1", FibonacciAir(), t, fibonacci_public_values(t), (1, 7, 8)
    yield "mul64", SynthMulAir(64), generate_synth_mul_trace(32), np.zeros(0, dtype=np.uint32), (2, 9, 8)
    yield "mul7_b3", SynthMulAir(7), generate_synth_mul_trace(16, 7), np.zeros(0, dtype=np.uint32), (3, 5, 4)
    yield "ext25", SynthExtAir(25), generate_synth_ext_trace(64, 25), np.zeros(0, dtype=np.uint32), (2, 6, 8)


def test_native_verifier_accepts_oracle_proofs_and_agrees_on_tampering(lib, orc):
    """uni_stark::verify shipped in the product (host C++), against the oracle's restatement:
    same verdict on valid proofs and on every single-word corruption tried."""
    rng = np.random.default_rng(7)
    for name, air, trace, pis, cfg in _cases():
        tape = ts.air_tape(air, len(pis))
        ocfg = orc.FriConfig(*cfg)
        proof = orc.prove(ocfg, tape, trace, pis)
        config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), host_only=True))
        chal = ts.BfChallenger()
        ts.verify(config, air, chal, proof, pis)  # Ok(())
        ochal = orc.OracleChallenger()
        assert orc.verify(ocfg, tape, proof, pis, ochal) == 0
        assert chal.state()[:16].tolist() == list(ochal.c.state), "verifier transcripts differ"
        positions = list(rng.integers(5, len(proof), size=40)) + [len(proof) - 1, len(proof) - 5, 5, 13]
        for pos in positions:
            bad = proof.copy()
            bad[pos] = (int(bad[pos]) + 1) % 0x78000001
            want = orc.verify(ocfg, tape, bad, pis)
            try:
                ts.verify(config, air, ts.BfChallenger(), bad, pis)
                got = 0
            except ts.VerificationError as e:
                got = e.code
            assert (got == 0) == (want == 0), (name, pos, got, want)
            if want != 0:
                assert got == want, (name, pos, got, want)
        # wrong public values / wrong query count / truncated
        if len(pis):
            wrong = pis.copy()
            wrong[-1] = (int(wrong[-1]) + 1) % 0x78000001
            with pytest.raises(ts.VerificationError) as ei:
                ts.verify(config, air, ts.BfChallenger(), proof, wrong)
            assert ei.value.code == 7
        other = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(cfg[0], cfg[1] + 1, cfg[2]), host_only=True))
        with pytest.raises(ts.VerificationError) as ei:
            ts.verify(other, air, ts.BfChallenger(), proof, pis)
        assert ei.value.code == 2
        with pytest.raises(ts.VerificationError) as ei:
            ts.verify(config, air, ts.BfChallenger(), proof[:-2], pis)
        assert ei.value.code == 9


def test_host_only_air_reports_degrees(lib, orc):
    for air, npub in ((FibonacciAir(), 3), (SynthMulAir(64), 0)):
        tape = ts.air_tape(air, npub)
        cair = ts.CompiledAir(None, tape)
        assert cair.log_quotient_degree == orc.log_quotient_degree(tape) == ts.get_log_quotient_degree(air, npub)
        assert cair.max_constraint_degree == orc.max_constraint_degree(tape)
        assert not cair.is_jit


# ------------------------------------------------------------------ proof wire format (postcard)
def _postcard_reference_encoder(proof: "ts.Proof") -> bytes:
    """postcard 1.0 by the book, written against the reference's struct definitions
    (uni-stark/src/proof.rs:19-38, fri/src/proof.rs:13-33, fri/src/two_adic_pcs.rs:63-68) from the
    PARSED proof -- independent of the C++ encoder, which walks the TSPF words."""
    out = bytearray()

    def varint(v):
        v = int(v)
        while v >= 0x80:
            out.append((v & 0x7F) | 0x80)
            v >>= 7
        out.append(v)

    def felts(a):
        for v in np.asarray(a).reshape(-1):
            varint(v)

    def digest(d):
        out.extend(np.asarray(d, dtype="<u4").tobytes())

    def commitment(d):  # Vec<[[u8; 4]; 8]> holding one root
        varint(1)
        digest(d)

    def path(p):  # Vec<[u8; 32]>
        varint(len(p))
        for d in p:
            digest(d)

    # Proof.commitments
    commitment(proof.trace_commit)
    commitment(proof.quotient_commit)
    # Proof.opened_values
    varint(len(proof.trace_local)); felts(proof.trace_local)
    varint(len(proof.trace_next)); felts(proof.trace_next)
    varint(len(proof.quotient_chunks))
    for ch in proof.quotient_chunks:
        varint(len(ch)); felts(ch)
    # Proof.opening_proof: FriProof
    varint(len(proof.commit_phase_commits))
    for c in proof.commit_phase_commits:
        commitment(c)
    varint(len(proof.query_proofs))
    for qp in proof.query_proofs:
        varint(len(qp.input_proof))
        for batch in qp.input_proof:
            varint(len(batch.opened_values))
            for row in batch.opened_values:
                varint(len(row)); felts(row)
            path(batch.opening_proof)
        varint(len(qp.commit_phase_openings))
        for vals, pth in qp.commit_phase_openings:
            varint(1); varint(2); felts(vals)
            path(pth)
    felts(proof.final_poly)
    # `type Witness = PF`, PF = [u8; 4] (basic/src/challenger/mod.rs:91, chan_field.rs:61): postcard
    # writes a fixed-size byte array as its raw bytes -- no length, no varint, no field reduction
    out.extend(int(proof.pow_witness).to_bytes(4, "little"))
    # Proof.degree_bits
    varint(proof.degree_bits)
    return bytes(out)


def test_postcard_wire_format_round_trip(lib, orc):
    for name, air, trace, pis, cfg in _cases():
        tape = ts.air_tape(air, len(pis))
        words = orc.prove(orc.FriConfig(*cfg), tape, trace, pis)
        proof = ts.Proof.parse(words)
        data = proof.to_postcard()
        assert data == _postcard_reference_encoder(proof), name
        back = ts.Proof.from_postcard(data)
        assert (back.words == words).all(), name
        # still a valid proof after the round trip
        config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), host_only=True))
        ts.verify(config, air, ts.BfChallenger(), back, pis)
        assert len(data) <= 5 * len(words)  # a 31-bit element takes up to 5 varint bytes
        # malformed inputs are refused, not mis-parsed
        for bad in (data[:-1], data + b"\x00", data[:40], b""):
            with pytest.raises(_lib.TsError):
                ts.Proof.from_postcard(bad)
        nc = bytearray(data)
        # first opened value -> a non-canonical element (>= p): 5-byte varint of 0xffffffff
        off = 2 * 33 + 1
        v = 0
        end = off
        while nc[end] & 0x80:
            end += 1
        nc[off:end + 1] = b"\xff\xff\xff\xff\x0f"
        with pytest.raises(_lib.TsError):
            ts.Proof.from_postcard(bytes(nc))
        # the witness is [u8; 4], not a field element: the 4 bytes before the trailing degree_bits
        # varint are its little-endian bytes, and any 32-bit value survives the round trip
        assert data[-5:-1] == int(proof.pow_witness).to_bytes(4, "little") and data[-1] == proof.degree_bits
        big = bytearray(data)
        big[-5:-1] = b"\xff\xff\xff\xff"
        assert ts.Proof.from_postcard(bytes(big)).pow_witness == 0xFFFFFFFF
        assert ts.Proof.from_postcard(bytes(big)).to_postcard() == bytes(big)
    with pytest.raises(_lib.TsError):
        ts.Proof(words=words[:10]).to_postcard()
    # a non-canonical opened value in the words is refused by the encoder too
    w2 = words.copy()
    w2[5 + 16] = 0x78000001
    with pytest.raises(_lib.TsError):
        ts.Proof(words=w2).to_postcard()
    # a ten-byte input that announces 2^24-element rows is refused at once (no half-gigabyte of zeros)
    import time
    t0 = time.time()
    with pytest.raises(_lib.TsError):
        ts.Proof.from_postcard(b"\x01" + bytes(32) + b"\x01" + bytes(32) + b"\x80\x80\x80\x08" + b"\x00" * 6)
    assert time.time() - t0 < 0.5


# ------------------------------------------------------------------ Pcs::verify, any shape (host)
PCS_VERIFY_SHAPES = [[[3]], [[2, 1]], [[2] * 5], [[4, 3]], [[3], [2]], [[2], [3, 3]], [[4, 2], [4, 2]],
                     [[0, 1], [2]], [[6, 3, 6], [5, 2]]]


@pytest.mark.parametrize("log_blowup", [1, 2])
@pytest.mark.parametrize("shape", PCS_VERIFY_SHAPES, ids=[str(s) for s in PCS_VERIFY_SHAPES])
def test_native_pcs_verify_on_oracle_openings(lib, orc, log_blowup, shape):
    # the second half of fri/tests/pcs.rs:62-117: what the (oracle) prover opened, the product's
    # Pcs::verify accepts from a fresh challenger in the same state -- and nothing else
    from tapstark_amd.airs import splitmix64_stream
    cfg = (log_blowup, 3, 8)
    seed, evals = 500, []
    for logs in shape:
        evs = []
        for lg in logs:
            seed += 1
            evs.append(splitmix64_stream(seed, (1 << lg) * (2 + seed % 3)).reshape(1 << lg, 2 + seed % 3))
        evals.append(evs)
    roots, zeta, opened, proof = orc.pcs_commit_open(orc.FriConfig(*cfg), shape, evals)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(*cfg), host_only=True)

    def claims(opened_vals, roots_=roots):
        out, k = [], 0
        for r, (logs, evs) in enumerate(zip(shape, evals)):
            mats = []
            for lg, e in zip(logs, evs):
                w = e.shape[1]
                mats.append((lg, [(zeta, opened_vals[k:k + w])]))
                k += w
            out.append((roots_[r], mats))
        return out

    def transcript():
        ch = ts.BfChallenger()
        for r in roots:
            ch.observe_commitment(r)
        assert (ch.sample() == zeta).all()
        return ch

    pcs.verify(claims(opened), proof, transcript())  # Ok(())
    # a wrong opened value, a wrong commitment, a corrupted proof word, a wrong transcript
    # (tampering with the LAST matrix: a height-1 matrix -- shape [0, ...], outside fri/tests/pcs.rs --
    # has an LDE of exactly `blowup` rows, whose reduced opening no FRI round consumes; the
    # reference only debug-asserts that case away, fri/src/verifier.rs:158-163)
    bad = opened.copy()
    bad[-1, 1] = (int(bad[-1, 1]) + 1) % 0x78000001
    with pytest.raises(ts.VerificationError):
        pcs.verify(claims(bad), proof, transcript())
    bad_roots = roots.copy()
    bad_roots[-1, 3] ^= 1
    with pytest.raises(ts.VerificationError) as ei:
        pcs.verify(claims(opened, bad_roots), proof, transcript())
    assert ei.value.code == 4
    for pos in (len(proof) - 1, len(proof) - 3, len(proof) // 2, 9):
        badp = proof.copy()
        badp[pos] = (int(badp[pos]) + 1) % 0x78000001
        with pytest.raises(ts.VerificationError):
            pcs.verify(claims(opened), badp, transcript())
    with pytest.raises(ts.VerificationError):
        pcs.verify(claims(opened), proof, ts.BfChallenger())
    with pytest.raises(ts.VerificationError) as ei:
        pcs.verify(claims(opened), proof[:-1], transcript())
    assert ei.value.code == 9


# ------------------------------------------------------------------ compiled-language host (C++)
def _build_example(tmp_path, name="fib_air"):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / name)
    libdir = os.path.join(root, "tap-stark_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-pthread", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", name + ".cpp"), "-L", libdir, "-ltapstark_hip",
                           f"-Wl,-rpath,{libdir}", "-o", exe])
    return exe


def test_cpp_stream_example_builds_with_plain_gxx(lib, tmp_path):
    # examples/prove_stream.cpp (several lanes, start gate, pinned uploads) links against the C ABI with
    # plain g++; without a GPU it must fail loudly at ts_ctx_create, not fall back
    import subprocess
    exe = _build_example(tmp_path, "prove_stream")
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "10", "4", "2"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "no MI355X context" in r.stderr


def test_cpp_sharded_example_builds_with_plain_gxx(lib, tmp_path):
    # examples/prove_sharded.cpp (G thread-ranks, in-process or native RCCL communicator, ts_prove_sharded)
    # links against the C ABI with plain g++; without a GPU it fails loudly at ts_ctx_create
    import subprocess
    exe = _build_example(tmp_path, "prove_sharded")
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "10", "2"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "no MI355X context" in r.stderr


def test_cpp_air_capture_matches_python(lib, tmp_path):
    """include/tapstark_air.hpp (the C++ SymbolicAirBuilder) captures FibonacciAir::eval into the
    very tape the Python front-end produces; the example links against the C ABI with plain g++."""
    import subprocess
    exe = _build_example(tmp_path)
    out = subprocess.run([exe, "--tape"], capture_output=True, text=True, check=True).stdout.splitlines()
    tape = np.array([int(x) for x in out[0].split()], dtype=np.uint32)
    assert (tape == ts.air_tape(FibonacciAir(), 3)).all()
    assert out[1].split() == [str(ts.get_max_constraint_degree(FibonacciAir(), 3)),
                              str(ts.get_log_quotient_degree(FibonacciAir(), 3))]


# ------------------------------------------------------------------ FRI alone (fri/tests/fri.rs)
def _fri_rs_inputs(orc, log_blowup, deg_bits_range, seed=0):
    """fri.rs:68-97: one random degree-2^k polynomial per k, LDE on the coset 31*H with bit-reversed
    rows, alpha = 1 and one column per matrix => the reduced opening of a height IS its LDE column;
    by descending height, embedded in EF4."""
    from tapstark_amd.airs import splitmix64_stream
    ins = []
    for k in deg_bits_range:
        ev = splitmix64_stream(seed + k, 1 << k).reshape(1 << k, 1)
        lde = orc.commit_lde(ev, 1, log_blowup)[:, 0]
        v = np.zeros((len(lde), 4), dtype=np.uint32)
        v[:, 0] = lde
        ins.append(v)
    return ins[::-1]


@pytest.mark.parametrize("perm,ext", [(1, False), (0, True)])
def test_native_fri_verify_on_oracle_fri_proofs(lib, orc, perm, ext):
    # test_compelte_fri_process (fri.rs:51-147): TestPermutation + BabyBear challenges as there, and
    # the production challenger; prover = oracle here (the GPU prover is checked in the gpu suite)
    cfg = (1, 10, 8)
    ins = _fri_rs_inputs(orc, cfg[0], range(1, 10))
    pch = orc.OracleChallenger(perm_kind=perm, sample_ext=ext)
    proof = orc.fri_prove(orc.FriConfig(*cfg), ins, pch)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(*cfg), host_only=True)
    vch = ts.BfChallenger(perm, ext)
    pcs.fri_verify(proof, vch)
    # fri.rs:141-146 "prover and verifier transcript have same state after FRI"
    assert vch.sample_bits(8) == pch.sample_bits(8)
    rejected = 0
    for pos in (len(proof) - 1, len(proof) - 2, 3, len(proof) // 2, len(proof) // 3):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % 0x78000001
        want = orc.fri_verify(orc.FriConfig(*cfg), bad, orc.OracleChallenger(perm_kind=perm, sample_ext=ext))
        try:
            pcs.fri_verify(bad, ts.BfChallenger(perm, ext))
            got = 0
        except ts.VerificationError as e:
            got = e.code
        assert got == want, (pos, got, want)
        rejected = rejected + (want != 0)
    assert rejected >= 3  # (a changed PoW witness can still be a witness under the test permutation)


@pytest.mark.parametrize("R", [26, 27, 31, 32, 0xFFFFFFFF])
def test_native_fri_verify_refuses_oversized_round_count(lib, R):
    # a caller-supplied proof may claim any round count: R + log_blowup > 27 (BabyBear's
    # two-adicity) must be refused before the query loop indexes its per-height arrays or the
    # challenger is asked for more than 27 bits (fri/src/verifier.rs:20-60 shape check)
    cfg = (2, 1, 0)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(*cfg), host_only=True)
    R_eff = min(R, 40)
    words = [R] + [0] * (8 * R_eff) + [1]  # Q = 1
    words += [0]  # no reduced openings in the input proof
    for r in range(R_eff):
        words += [0] * 8 + [0]
    words += [0, 0, 0, 0, 0]
    with pytest.raises(ts.VerificationError):
        pcs.fri_verify(np.array(words, dtype=np.uint32), ts.BfChallenger())


def test_native_pcs_verify_rows_wider_than_one_blake3_chunk(lib, orc):
    # bf_mmcs.rs:17-68 takes matrices of any width; a row of more than 256 elements is more than
    # one Blake3 chunk (chunk chaining + parent nodes).  Prover = oracle, verifier = the product's
    # host code (csrc/blake3.hpp hash_stream); the widths straddle 1, 2 and 3 chunks.
    from tapstark_amd.airs import splitmix64_stream
    cfg = (1, 2, 4)
    shape = [[3, 2]]
    evals = [[splitmix64_stream(91, 8 * 300).reshape(8, 300), splitmix64_stream(92, 4 * 520).reshape(4, 520)]]
    roots, zeta, opened, proof = orc.pcs_commit_open(orc.FriConfig(*cfg), shape, evals)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(*cfg), host_only=True)

    def transcript():
        ch = ts.BfChallenger()
        ch.observe_commitment(roots[0])
        assert (ch.sample() == zeta).all()
        return ch

    claims = [(roots[0], [(3, [(zeta, opened[:300])]), (2, [(zeta, opened[300:820])])])]
    pcs.verify(claims, proof, transcript())
    bad = proof.copy()
    # a word inside the first opened row of the first query (R, commits, Q, n_batches, n_mats, width, row...)
    pos = 1 + 8 * int(proof[0]) + 1 + 3 + 299
    bad[pos] = (int(bad[pos]) + 1) % 0x78000001
    with pytest.raises(ts.VerificationError):
        pcs.verify(claims, bad, transcript())
