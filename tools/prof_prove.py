"""Runs a few whole proofs of a BASELINE config (for rocprofv3 --kernel-trace --stats / --pmc):
    python tools/prof_prove.py [n_proofs] [config3|config2|config4|config5]
The trace is generated on the device; one untimed proof first builds the per-context tables."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
from bench import workload

n_proofs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
name = sys.argv[2] if len(sys.argv) > 2 else "config3"
ctx = ts.default_context()
air, _, pis, desc, cfg, (n, w), make_trace = workload(name, 22 if name == "config4" else 20, False)
if callable(pis):
    pis = np.array([0, 1, pis(ctx)], dtype=np.uint32)
config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
for _ in range(n_proofs):
    p = ts.prove(config, cair, ts.BfChallenger(), make_trace(ctx), pis)
print("ok", desc, len(p.words))
