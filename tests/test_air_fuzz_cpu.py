"""Fuzz of the AIR front half on the CPU (no GPU): seeded random AIRs (tap-stark_amd/airs.py
RandomAir) through the product's tape validation, degree rules, lowering to the register program
and host verifier, against the oracle.  The reference's prove() is generic over `Air`
(uni-stark/src/prover.rs:25-39, symbolic_builder.rs:15-64, symbolic_expression.rs:41-61,137,182,227,
folder.rs:44-64); the GPU half (quotient kernels, whole proofs) is tests/test_gpu_air_fuzz.py.
TS_AIR_FUZZ=<n> widens the campaign."""
import os

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd.airs import (RandomAir, generate_random_air_trace, random_air_case, splitmix64_stream,
                               NumericBuilder)
from _air_program import run_program

P = 0x78000001
N_CASES = int(os.environ.get("TS_AIR_FUZZ", "300"))


def _inputs(seed, w, m=6):
    """m (local, next, selectors) inputs with edge values mixed in."""
    vals = splitmix64_stream(seed + 77, 2 * m * w + 3 * m)
    local = vals[:m * w].reshape(m, w).copy()
    nxt = vals[m * w:2 * m * w].reshape(m, w).copy()
    sels = vals[2 * m * w:].reshape(m, 3).copy()
    local[0, :] = 0
    nxt[0, :] = P - 1
    local[1, :] = P - 1
    sels[0] = (1, 0, 1)  # check_constraints' first row
    sels[1] = (0, 1, 0)  # ... and last row
    sels[2] = (0, 0, 1)
    return local, nxt, sels


@pytest.mark.parametrize("chunk", range(10))
def test_register_program_matches_oracle(orc, chunk):
    per = (N_CASES + 9) // 10
    for seed in range(chunk * per, (chunk + 1) * per):
        air, _ = random_air_case(seed)
        tape = ts.air_tape(air, air.n_public)
        assert orc.tape_validate(tape) == 0
        cair = ts.CompiledAir(None, tape)
        assert cair.max_constraint_degree == orc.max_constraint_degree(tape) == air.max_degree, seed
        assert cair.log_quotient_degree == orc.log_quotient_degree(tape), seed
        assert cair.log_quotient_degree == ts.get_log_quotient_degree(air, air.n_public), seed
        local, nxt, sels = _inputs(seed, air.width())
        pis = splitmix64_stream(seed + 5, max(air.n_public, 1))[:air.n_public]
        want = orc.constraint_values(tape, local, nxt, pis, sels)
        prog = cair.program()
        got = run_program(prog, local, nxt, pis, sels, int(tape[5]))
        assert (got == want).all(), f"seed {seed}: constraint values differ"
        # a register is only worth having if something lives in it
        assert prog["n_regs"] <= max(1, len(prog["code"]))


@pytest.mark.parametrize("seed", range(0, 120, 3))
def test_valid_random_air_is_proved_and_verified(orc, seed):
    """valid=True cases: the generated trace satisfies the AIR (numpy check, oracle check_constraints),
    the oracle's proof is accepted by the oracle's verifier AND by the product's host verifier
    (csrc/verifier.cpp evaluates the same tape at zeta in EF4), and a corrupted cell is reported at
    the same (row, constraint) by the numpy evaluation and the oracle."""
    air, log_n = random_air_case(seed)
    if air.width() * air.n_constraints > 20000:
        log_n = min(log_n, 2)
    tape = ts.air_tape(air, air.n_public)
    trace, pis, nb = generate_random_air_trace(air, 1 << log_n)
    assert nb.first_violation() == -1 == orc.check_constraints(tape, trace, pis)
    lqd = orc.log_quotient_degree(tape)
    cfg = (max(lqd, 1), 4, 2)
    proof = orc.prove(orc.FriConfig(*cfg), tape, trace, pis)
    assert orc.verify(orc.FriConfig(*cfg), tape, proof, pis) == 0
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), None, host_only=True))
    ts.verify(config, ts.CompiledAir(None, tape), ts.BfChallenger(), proof, pis)
    bad = trace.copy()
    bad[(seed * 7) % len(bad), seed % air.width()] ^= 1
    nbad = NumericBuilder(bad.astype(np.uint64), pis, define=False)
    air.eval(nbad)
    assert nbad.first_violation() == orc.check_constraints(tape, bad, pis)


def test_large_tape_lowering(orc):
    """>= 10^4 nodes: lowering stays linear-time and the program still matches the oracle."""
    import time

    air = RandomAir(4242, 200, 3000, 5, n_public=4, share_pct=20, max_depth=7)
    tape = ts.air_tape(air, 4)
    assert int(tape[4]) >= 10_000
    t0 = time.time()
    cair = ts.CompiledAir(None, tape)
    assert time.time() - t0 < 5.0
    local, nxt, sels = _inputs(1, 200, m=4)
    pis = splitmix64_stream(9, 4)
    want = orc.constraint_values(tape, local, nxt, pis, sels)
    got = run_program(cair.program(), local, nxt, pis, sels, int(tape[5]))
    assert (got == want).all()


def test_code_object_cache(orc, tmp_path, monkeypatch):
    """TS_JIT_CACHE_DIR: the code object of a source is compiled once and then read back (no GPU needed:
    hiprtc cross-compiles); a damaged file is not trusted."""
    import glob
    import time

    air = RandomAir(11, 40, 60, 3)
    cair = ts.CompiledAir(None, ts.air_tape(air, 3))
    monkeypatch.setenv("TS_JIT_CACHE_DIR", str(tmp_path))
    try:
        code1, t1 = cair.jit_compile()
    except ts._lib.TsError as e:  # no libhiprtc in this environment
        pytest.skip(str(e))
    files = glob.glob(str(tmp_path / "q_*_gfx950.co"))
    assert len(files) == 1 and open(files[0], "rb").read() == code1
    t0 = time.time()
    code2, _ = cair.jit_compile()
    assert code2 == code1 and time.time() - t0 < 0.5 * t1 + 0.05
    other = ts.CompiledAir(None, ts.air_tape(RandomAir(12, 40, 60, 3), 3))
    other.jit_compile()
    assert len(glob.glob(str(tmp_path / "q_*_gfx950.co"))) == 2  # another source, another entry
    open(files[0], "wb").write(b"not an ELF file")
    code3, _ = cair.jit_compile()
    assert code3[:4] == b"\x7fELF" and len(code3) == len(code1)
