"""Runs a few whole proofs of the BASELINE config-3 shape (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
from tapstark_amd.airs import SynthMulAir, generate_synth_mul_trace

ctx = ts.default_context()
config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 28, 8), ctx))
air = SynthMulAir(64)
cair = ts.CompiledAir(ctx, ts.air_tape(air, 0))
trace = generate_synth_mul_trace(1 << 20)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    p = ts.prove(config, cair, ts.BfChallenger(), trace, [])
print("ok", len(p.words))
