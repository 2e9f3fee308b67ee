"""Pins the BASELINE.json configurations at their FULL sizes: runs the CPU oracle's prove() on
configs 2, 3, 4 and 5 and writes tests/golden/large_fixtures.json -- per config the Blake3 of the
trace, of the proof words, of the opened values, and every commitment root (trace, quotient, each
FRI round) plus the transcript's challenges, so that a mismatch of a GPU proof localises to a stage
(SURVEY.md section 8(c) "seeds + digests ... for the large configs").

The reference (uni-stark/src/prover.rs:25-119) is Rust and cannot run here; the oracle is the
restatement tests/test_oracle.py pins.  Run in the build container (8 cores, 62 GiB):

    python tests/golden/make_golden_large.py config2 config3 config5     # ~1, ~1, ~5 min
    python tests/golden/make_golden_large.py config4                     # ~10 min, ~40 GiB peak
    python tests/golden/make_golden_large.py fib_2p24_b2 mul64_2p23_b1 fib_2p25_b1 fib_2p26_b1

Each run merges its configs into the existing JSON.  The buffers themselves are far too large to
commit (config 4's LDE is 16 GiB); the digests are what travels.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import tapstark_amd as ts  # noqa: E402  (host-only use: the AIR tape serialiser and trace generators)
from _digests import digest_record  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
from tapstark_amd.airs import (FibonacciAir, SynthExtAir, SynthMulAir, fibonacci_public_values,  # noqa: E402
                               generate_fibonacci_trace, generate_synth_ext_trace,
                               generate_synth_mul_trace)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "large_fixtures.json")

# name -> (air, trace generator, public values from the trace, (log_blowup, queries, pow bits), log_n)
CONFIGS = {
    "config2": (FibonacciAir, lambda n: generate_fibonacci_trace(0, 1, n), fibonacci_public_values, (2, 28, 8), 20),
    "config3": (lambda: SynthMulAir(64), generate_synth_mul_trace, None, (2, 28, 8), 20),
    "config4": (lambda: SynthMulAir(64), generate_synth_mul_trace, None, (4, 16, 8), 22),
    "config5": (lambda: SynthExtAir(163), lambda n: generate_synth_ext_trace(n, 163), None, (4, 16, 8), 20),
    # beyond the BASELINE sizes: the LDE plans for n = 2^23 .. 2^26 (strided pass of 11 .. 14 stages) up to
    # the largest LDE the field allows (2^27 rows): ~100 s, ~100 s and ~4 min / 35 GiB of oracle time
    "fib_2p24_b2": (FibonacciAir, lambda n: generate_fibonacci_trace(0, 1, n), fibonacci_public_values, (2, 28, 8), 24),
    "fib_2p25_b1": (FibonacciAir, lambda n: generate_fibonacci_trace(0, 1, n), fibonacci_public_values, (1, 28, 8), 25),
    "fib_2p26_b1": (FibonacciAir, lambda n: generate_fibonacci_trace(0, 1, n), fibonacci_public_values, (1, 28, 8), 26),
    "mul64_2p23_b1": (lambda: SynthMulAir(64), generate_synth_mul_trace, None, (1, 16, 8), 23),
}


def make(name: str) -> dict:
    mk_air, gen, pis_of, cfg, log_n = CONFIGS[name]
    air = mk_air()
    t0 = time.time()
    trace = gen(1 << log_n)
    pis = pis_of(trace) if pis_of else np.zeros(0, dtype=np.uint32)
    tape = ts.air_tape(air, len(pis))
    rec = {"air": type(air).__name__, "log_n": log_n, "width": int(trace.shape[1]),
           "log_blowup": cfg[0], "num_queries": cfg[1], "proof_of_work_bits": cfg[2],
           "public_values": [int(x) for x in pis],
           "trace_blake3": orc.blake3(trace.tobytes()).hex()}
    print(f"{name}: trace {trace.shape} ready after {time.time() - t0:.1f} s", flush=True)
    t1 = time.time()
    ocfg = orc.FriConfig(*cfg)
    proof = orc.prove(ocfg, tape, trace, pis)
    rec["oracle_prove_seconds_here"] = round(time.time() - t1, 1)
    print(f"{name}: oracle prove {rec['oracle_prove_seconds_here']} s, {len(proof)} words", flush=True)
    assert orc.verify(ocfg, tape, proof, pis) == 0, "oracle verifier rejects its own proof"
    tr = orc.last_transcript()
    rec.update(digest_record(proof))
    rec["alpha"] = [int(x) for x in tr["alpha"]]
    rec["zeta"] = [int(x) for x in tr["zeta"]]
    rec["batch_alpha"] = [int(x) for x in tr["batch_alpha"]]
    rec["betas"] = [[int(x) for x in b] for b in tr["betas"]]
    rec["query_indices"] = [int(x) for x in tr["indices"]]
    return rec


if __name__ == "__main__":
    names = sys.argv[1:] or ["config2", "config3", "config5"]
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for nm in names:
        out[nm] = make(nm)
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        print(f"{nm}: written to {OUT}", flush=True)
