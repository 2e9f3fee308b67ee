"""Regenerates tests/golden/oracle_fixtures.json from the CPU oracle (the reference is Rust and
cannot be run here, so there is nothing to import).  The fixtures freeze
the oracle's own outputs so that later edits to it cannot drift silently; the reference-published
known answers live in kats.json.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import tapstark_amd as ts  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402
from tapstark_amd.airs import (FibonacciAir, SynthMulAir, fibonacci_public_values,  # noqa: E402
                               generate_fibonacci_trace, generate_synth_mul_trace)

out = {}
out["blake3_pattern"] = {
    str(ln): orc.blake3(bytes((i * 7 + 3) & 0xFF for i in range(ln))).hex()
    for ln in [0, 1, 63, 64, 65, 128, 256, 652, 1023, 1024, 1025, 2048, 3072, 5000]
}

air = FibonacciAir()
trace = generate_fibonacci_trace(0, 1, 8)
pis = fibonacci_public_values(trace)
tape = ts.air_tape(air, 3)
cfg = orc.FriConfig(2, 28, 8)
proof = orc.prove(cfg, tape, trace, pis)
assert orc.verify(cfg, tape, proof, pis) == 0
tr = orc.last_transcript()
out["fib8_q28_proof_blake3"] = orc.blake3(proof.tobytes()).hex()
out["fib8_q28_proof_words"] = int(len(proof))
out["fib8_q28_alpha"] = tr["alpha"].tolist()
out["fib8_q28_zeta"] = tr["zeta"].tolist()
out["fib8_q28_batch_alpha"] = tr["batch_alpha"].tolist()
out["fib8_q28_betas"] = tr["betas"].tolist()
out["fib8_q28_pow_witness"] = tr["pow_witness"]
out["fib8_q28_indices"] = tr["indices"].tolist()
lde = orc.commit_lde(trace, 1, 2)
out["fib8_trace_lde_bitrev"] = lde.tolist()
out["fib8_trace_root"] = orc.OracleMmcs([lde]).root.tolist()
q = orc.quotient_values(tape, lde, 3, 2, pis, tr["alpha"])
out["fib8_quotient_values"] = q.tolist()

# seeds + digests for a mid-size wide case (buffers themselves are too large to commit)
air = SynthMulAir(64)
tape = ts.air_tape(air, 0)
trace = generate_synth_mul_trace(1 << 10)
cfg = orc.FriConfig(2, 28, 8)
proof = orc.prove(cfg, tape, trace, [])
assert orc.verify(cfg, tape, proof, []) == 0
out["synthmul64_2pow10_trace_blake3"] = orc.blake3(trace.tobytes()).hex()
out["synthmul64_2pow10_proof_blake3"] = orc.blake3(proof.tobytes()).hex()
out["synthmul64_2pow10_lde_blake3"] = orc.blake3(orc.commit_lde(trace, 1, 2).tobytes()).hex()

with open(os.path.join(os.path.dirname(__file__), "oracle_fixtures.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote oracle_fixtures.json")
