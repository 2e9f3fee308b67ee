"""Fuzz of the one user-programmable stage on the GPU: seeded random AIRs (tap-stark_amd/airs.py
RandomAir: shared sub-terms, long live ranges, selectors inside and outside products, public values
in high-degree terms, edge constants, degrees 1..9, widths 1..200, up to 3000 constraints) through
both constraint compilers -- the hiprtc-specialised kernel (csrc/jit.cpp) and the on-device
interpreter (csrc/quotient.hip) -- against the oracle's direct evaluation of the tape:

* degree rules            vs orc.max_constraint_degree / log_quotient_degree
                          (uni-stark/src/symbolic_builder.rs:15-64, symbolic_expression.rs:41-61,137,182,227)
* quotient_chunks         vs orc.quotient_values + split  (prover.rs:122-194, folder.rs:44-64)
* check_constraints       vs orc.check_constraints        (check_constraints.rs:11-39)
* whole proofs (a subset) vs orc.prove, word for word; valid-trace cases also verified

Default: 240 seeds (TS_AIR_FUZZ=<n> [TS_AIR_FUZZ_FIRST=<seed>] for a campaign slice; the summary goes
to gpurun_out/air_fuzz/)."""
import json
import os
import time

import numpy as np
import pytest

import tapstark_amd as ts
from tapstark_amd.airs import (NumericBuilder, RandomAir, generate_random_air_trace, random_air_case,
                               splitmix64_stream)

pytestmark = pytest.mark.gpu
P = 0x78000001
N_CASES = int(os.environ.get("TS_AIR_FUZZ", "240"))
N_CHUNKS = 20
# background-compiled programs are waited for (and the specialised kernel compared) up to this size
WAIT_JIT_INSTR = int(os.environ.get("TS_AIR_FUZZ_WAIT_INSTR", "3000"))
SUMMARY = {"cases": 0, "valid_cases": 0, "jit_compared": 0, "interp_compared": 0, "proofs_compared": 0,
           "check_constraints_compared": 0, "background_jit": 0, "jit_not_waited": 0, "mismatches": [],
           "refusals": [], "max_nodes": 0, "max_constraints": 0, "max_regs": 0, "max_instr": 0,
           "by_degree": {}, "jit_compile_s_max": 0.0, "jit_compile_s_total": 0.0, "phase_s": {}}
FIRST = int(os.environ.get("TS_AIR_FUZZ_FIRST", "0"))  # campaigns are run in slices: seeds FIRST .. FIRST + N


class _Phase:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t0 = time.time()

    def __exit__(self, *a):
        SUMMARY["phase_s"][self.name] = round(SUMMARY["phase_s"].get(self.name, 0.0) + time.time() - self.t0, 3)


@pytest.fixture(scope="module")
def ctx():
    from tapstark_amd.build import build

    build()
    return ts.default_context()


@pytest.fixture(scope="module", autouse=True)
def _write_summary():
    t0 = time.time()
    yield
    SUMMARY["seconds"] = round(time.time() - t0, 1)
    SUMMARY["n_requested"] = N_CASES
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "air_fuzz")
    os.makedirs(out, exist_ok=True)
    SUMMARY["first_seed"] = FIRST
    with open(os.path.join(out, f"summary_{FIRST}_{N_CASES}.json"), "w") as f:
        json.dump(SUMMARY, f, indent=1)


def _compile(ctx, tape, monkeypatch, jit: bool):
    # programs above the size this run waits for are left to the interpreter (TS_JIT_MAX_INSTR), like any
    # program above the product's own budget: a compiler child nobody waits for would only burn host cores
    with monkeypatch.context() as m:
        if not jit:
            m.setenv("TS_NO_JIT", "1")
        m.setenv("TS_JIT_MAX_INSTR", str(WAIT_JIT_INSTR))
        return ts.CompiledAir(ctx, tape)


def _check_chunks(pcs, data, cair, pis, alpha, want, what, seed):
    chunks = pcs.quotient_chunks(data, cair, pis, alpha)
    assert len(chunks) == want.shape[0], (seed, what)
    for c, ch in enumerate(chunks):
        got = ch.download()
        if not (got == want[c]).all():
            SUMMARY["mismatches"].append({"seed": seed, "path": what, "chunk": c})
            raise AssertionError(f"seed {seed} ({what}): chunk {c}: {int((got != want[c]).sum())} words differ")


def run_case(ctx, orc, monkeypatch, seed: int):
    with _Phase("generate"):
        air, log_n = random_air_case(seed)
        n = 1 << log_n
        tape = ts.air_tape(air, air.n_public)
        n_nodes, n_cons = int(tape[4]), int(tape[5])
        if air.valid:
            trace, pis, _ = generate_random_air_trace(air, n)
        else:
            trace = splitmix64_stream(seed + 1, n * air.width()).reshape(n, air.width())
            pis = splitmix64_stream(seed + 2, max(air.n_public, 1))[:air.n_public]
    t0 = time.time()
    with _Phase("ts_air_compile(jit)"):
        cair = _compile(ctx, tape, monkeypatch, True)
    dt = time.time() - t0
    with _Phase("ts_air_compile(interp)"):
        interp = _compile(ctx, tape, monkeypatch, False)
    assert not interp.is_jit
    prog = cair.program()
    S = SUMMARY
    S["cases"] += 1
    S["valid_cases"] += int(air.valid)
    S["max_nodes"], S["max_constraints"] = max(S["max_nodes"], n_nodes), max(S["max_constraints"], n_cons)
    S["max_regs"], S["max_instr"] = max(S["max_regs"], prog["n_regs"]), max(S["max_instr"], len(prog["code"]))
    S["by_degree"][str(air.max_degree)] = S["by_degree"].get(str(air.max_degree), 0) + 1
    # degree rules
    assert cair.max_constraint_degree == orc.max_constraint_degree(tape) == air.max_degree, seed
    lqd = orc.log_quotient_degree(tape)
    assert cair.log_quotient_degree == lqd == ts.get_log_quotient_degree(air, air.n_public), seed
    # quotient values on the committed LDE, both kernels
    b = max(lqd, 1)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 3, 2), ctx)
    _, data = pcs.commit([((log_n, 1), trace.copy())])
    alpha = splitmix64_stream(seed + 3, 4)
    with _Phase("oracle quotient"):
        lde = orc.commit_lde(trace, 1, b)
        want = orc.split_quotient(orc.quotient_values(tape, lde, log_n, b, pis, alpha), log_n, lqd)
    with _Phase("gpu quotient"):
        _check_chunks(pcs, data, interp, pis, alpha, want, "interp", seed)
    S["interp_compared"] += 1
    if not cair.is_jit:
        if len(prog["code"]) <= WAIT_JIT_INSTR:
            S["background_jit"] += 1
            with _Phase("jit_wait"):
                state, secs = cair.jit_wait()
            assert state == 3 and cair.is_jit, f"seed {seed}: background specialisation failed (state {state})"
            dt = secs
        else:
            S["jit_not_waited"] += 1  # above this run's budget: interpreter only
            assert cair.jit_wait()[0] == 0
    if cair.is_jit:
        S["jit_compile_s_max"] = max(S["jit_compile_s_max"], round(dt, 2))
        S["jit_compile_s_total"] = round(S["jit_compile_s_total"] + dt, 2)
        with _Phase("gpu quotient"):
            _check_chunks(pcs, data, cair, pis, alpha, want, "jit", seed)
        S["jit_compared"] += 1
    # check_constraints: the trace as it is, and with one cell changed
    with _Phase("check_constraints"):
        got = ts.check_constraints(interp, trace, pis, ctx)
        assert got == orc.check_constraints(tape, trace, pis), seed
        if air.valid:
            assert got == -1, seed
        bad = trace.copy()
        bad[(seed * 7) % n, seed % air.width()] ^= 1
        assert ts.check_constraints(interp, bad, pis, ctx) == orc.check_constraints(tape, bad, pis), seed
    S["check_constraints_compared"] += 2
    # whole proofs
    if seed % 4 == 0 or air.valid and seed % 2 == 0:
        cfg = (b, 3, 2)
        config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
        with _Phase("oracle prove"):
            ref = orc.prove(orc.FriConfig(*cfg), tape, trace, pis, debug_assertions=air.valid)
        for which in (cair, interp):
            with _Phase("gpu prove"):
                proof = ts.prove(config, which, ts.BfChallenger(), trace.copy(), pis)
            assert len(ref) == len(proof.words) and (ref == proof.words).all(), f"seed {seed}: proof differs"
            S["proofs_compared"] += 1
        rc = orc.verify(orc.FriConfig(*cfg), tape, proof.words, pis)
        assert rc == (0 if air.valid else 7), (seed, rc)
        if air.valid:
            ts.verify(config, cair, ts.BfChallenger(), proof, pis)


@pytest.mark.parametrize("chunk", range(N_CHUNKS))
def test_random_airs(ctx, orc, monkeypatch, chunk):
    per = (N_CASES + N_CHUNKS - 1) // N_CHUNKS
    for seed in range(chunk * per, min((chunk + 1) * per, N_CASES)):
        run_case(ctx, orc, monkeypatch, FIRST + seed)


def _resources(code: bytes) -> dict:
    """VGPRs / scratch of the kernel inside a code object (llvm-readelf --notes of the ROCm toolchain)."""
    import re
    import subprocess
    import tempfile

    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code)
        f.flush()
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True,
                             text=True).stdout
    res = {}
    for key in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count"):
        m = re.search(r"\." + key + r":\s+(\d+)", out)
        if m:
            res[key] = int(m.group(1))
    return res


def test_large_tape(ctx, orc, monkeypatch):
    """A tape of >= 10^4 nodes (3000 constraints over 200 columns, ~1500 values live at once):
    * the interpreter runs it (register file in a global slab: more live values than LDS holds),
    * its specialisation is compiled in the background and adopted when ready (compile seconds,
      VGPRs and scratch recorded), both match the oracle, and so does a whole proof."""
    air = RandomAir(4242, 200, 3000, 5, n_public=4, share_pct=20, max_depth=7)
    tape = ts.air_tape(air, 4)
    assert int(tape[4]) >= 10_000
    log_n, b = 6, 2
    n = 1 << log_n
    trace = splitmix64_stream(99, n * 200).reshape(n, 200)
    pis = splitmix64_stream(98, 4)
    large_jit = os.environ.get("TS_AIR_FUZZ_LARGE_JIT", "0") != "0"  # ~60 s of hiprtc: campaign runs only
    t0 = time.time()
    with monkeypatch.context() as m:
        if not large_jit:
            m.setenv("TS_JIT_MAX_INSTR", "0")
        cair = ts.CompiledAir(ctx, tape)
    t_compile_call = time.time() - t0
    prog = cair.program()
    rec = {"nodes": int(tape[4]), "constraints": int(tape[5]), "n_regs": prog["n_regs"],
           "n_instr": len(prog["code"]), "ts_air_compile_s": round(t_compile_call, 3)}
    assert t_compile_call < 10 and not cair.is_jit, "a program this size must not be compiled synchronously"
    assert prog["n_regs"] * 64 * 4 > 160 * 1024, "meant to exceed the LDS register file"
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 3, 2), ctx)
    _, data = pcs.commit([((log_n, 1), trace.copy())])
    alpha = splitmix64_stream(97, 4)
    lqd = orc.log_quotient_degree(tape)
    lde = orc.commit_lde(trace, 1, b)
    want = orc.split_quotient(orc.quotient_values(tape, lde, log_n, b, pis, alpha), log_n, lqd)
    ctx.synchronize()
    t0 = time.time()
    _check_chunks(pcs, data, cair, pis, alpha, want, "interp-global-regs", 4242)
    rec["interp_quotient_s"] = round(time.time() - t0, 4)
    assert ts.check_constraints(cair, trace, pis, ctx) == orc.check_constraints(tape, trace, pis)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(b, 3, 2), ctx))
    ref = orc.prove(orc.FriConfig(b, 3, 2), tape, trace, pis, debug_assertions=False)
    proof = ts.prove(config, cair, ts.BfChallenger(), trace.copy(), pis)
    assert (ref == proof.words).all()
    if large_jit:
        state, secs = cair.jit_wait()
        rec["jit_state"], rec["hiprtc_compile_s"] = state, round(secs, 1)
        assert state == 3 and cair.is_jit
        t0 = time.time()
        _check_chunks(pcs, data, cair, pis, alpha, want, "jit-large", 4242)
        rec["jit_quotient_s"] = round(time.time() - t0, 4)
        proof = ts.prove(config, cair, ts.BfChallenger(), trace.copy(), pis)
        assert (ref == proof.words).all()
        rec["jit_kernel"] = _resources(cair.jit_compile()[0]) if os.environ.get("TS_AIR_FUZZ_LARGE_RES") else None
    SUMMARY["large_tape"] = rec


def test_jit_budget_knobs(ctx, orc, monkeypatch):
    """TS_JIT_MAX_INSTR below the program: interpreter only, stated in is_jit; TS_JIT_SYNC_INSTR above
    it: compiled inside ts_air_compile."""
    air = RandomAir(7, 20, 40, 3)
    tape = ts.air_tape(air, 3)
    with monkeypatch.context() as m:
        m.setenv("TS_JIT_SYNC_INSTR", "1")
        m.setenv("TS_JIT_MAX_INSTR", "2")
        c = ts.CompiledAir(ctx, tape)
        assert not c.is_jit and c.jit_wait()[0] == 0
    with monkeypatch.context() as m:
        m.setenv("TS_JIT_SYNC_INSTR", "1")
        c = ts.CompiledAir(ctx, tape)  # background
        assert c.jit_wait()[0] == 3 and c.is_jit
    assert ts.CompiledAir(ctx, tape).is_jit


def test_code_object_cache_on_the_device(ctx, orc, monkeypatch, tmp_path):
    """TS_JIT_CACHE_DIR: the first ts_air_compile leaves the code object, a later one of the same AIR --
    inside or above the synchronous budget -- loads it without compiling, and computes the same chunks."""
    import glob

    air = RandomAir(21, 30, 50, 3)
    tape = ts.air_tape(air, 3)
    log_n, b = 6, 1
    trace = splitmix64_stream(5, (1 << log_n) * 30).reshape(1 << log_n, 30)
    pis, alpha = splitmix64_stream(6, 3), splitmix64_stream(7, 4)
    lqd = orc.log_quotient_degree(tape)
    b = max(lqd, 1)
    want = orc.split_quotient(orc.quotient_values(tape, orc.commit_lde(trace, 1, b), log_n, b, pis, alpha), log_n, lqd)
    pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 3, 2), ctx)
    _, data = pcs.commit([((log_n, 1), trace.copy())])
    monkeypatch.setenv("TS_JIT_CACHE_DIR", str(tmp_path))
    first = ts.CompiledAir(ctx, tape)
    assert first.is_jit and len(glob.glob(str(tmp_path / "q_*.co"))) == 1
    t0 = time.time()
    again = ts.CompiledAir(ctx, tape)
    assert again.is_jit and time.time() - t0 < 0.2, "a cached code object is loaded, not compiled"
    with monkeypatch.context() as m:
        m.setenv("TS_JIT_SYNC_INSTR", "1")  # the background route: the cache answers before a child is started
        bg = ts.CompiledAir(ctx, tape)
        assert bg.is_jit and bg.jit_wait()[0] == 3
    for c in (first, again, bg):
        _check_chunks(pcs, data, c, pis, alpha, want, "cached", 21)


# ------------------------------------------------------------------ random AIRs through the sharded prover
def _thread_ranks(G, rank_fn):
    import threading

    out, errors = [None] * G, [None] * G

    def main(r):
        try:
            out[r] = rank_fn(r)
        except BaseException as e:  # noqa: BLE001
            errors[r] = e

    threads = [threading.Thread(target=main, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank is stuck in a collective"
    for r in range(G):
        assert errors[r] is None, f"rank {r}: {errors[r]!r}"
    return out


SHARDED_SEEDS = ([0, 3, 6, 9, 12, 15, 18, 21, 24, 27, 1, 4] if "TS_AIR_FUZZ_SHARDED" not in os.environ
                 else list(range(int(os.environ["TS_AIR_FUZZ_SHARDED"]))))


@pytest.mark.parametrize("seed", SHARDED_SEEDS)
def test_random_airs_sharded(ctx, orc, monkeypatch, seed):
    """The same random AIRs as ONE proof over G thread-ranks (csrc/sharded.cpp): the quotient is then
    evaluated on row ranges / on each rank's own cosets with their shifts and mixed back
    (local_quotient), a different consumer of the lowered program.  Valid-trace seeds (multiples of 3)
    must give ts_prove's = the oracle's proof on both paths; the free-form ones (invalid trace) go
    through the local path's fall-back."""
    from tapstark_amd.comm import LocalCommGroup

    air, log_n = random_air_case(seed)
    log_n = max(log_n, 5)
    n = 1 << log_n
    tape = ts.air_tape(air, air.n_public)
    if air.valid:
        trace, pis, _ = generate_random_air_trace(air, n)
    else:
        trace = splitmix64_stream(seed + 1, n * air.width()).reshape(n, air.width())
        pis = splitmix64_stream(seed + 2, max(air.n_public, 1))[:air.n_public]
    lqd = orc.log_quotient_degree(tape)
    for b, G in ((max(lqd, 1), 2), (max(lqd, 1) + 1, 4)):
        cfg = (b, 4, 3)
        want = orc.prove(orc.FriConfig(*cfg), tape, trace, pis, debug_assertions=False)
        # rank 0 runs the specialised kernel (one compilation per configuration), the others the
        # interpreter: both take the row ranges / coset shifts of the sharded quotient, and the ranks'
        # slabs must still fit together into the oracle's proof
        ctxs = [ts.Context(0) for _ in range(G)]
        airs = [_compile(ctxs[r], tape, monkeypatch, r == 0) for r in range(G)]
        for localq in (False, True):
            group = LocalCommGroup(G)

            def rank(r):
                conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctxs[r]))
                rows = np.ascontiguousarray(trace[r * n // G:(r + 1) * n // G])
                return ts.prove_sharded(conf, airs[r], ts.BfChallenger(), rows, pis, group.comm(r),
                                        min_local_log=2, local_quotient=localq).words

            for r, words in enumerate(_thread_ranks(G, rank)):
                assert len(words) == len(want) and (words == want).all(), \
                    f"seed {seed} G={G} b={b} localq={localq} rank {r}: proof differs from the oracle's"
            SUMMARY["sharded_proofs_compared"] = SUMMARY.get("sharded_proofs_compared", 0) + G
