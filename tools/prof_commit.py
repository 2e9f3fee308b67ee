"""Runs a few trace commits (LDE + Merkle) of the BASELINE config-3 shape, for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
from tapstark_amd.airs import generate_synth_mul_trace

ctx = ts.default_context()
pcs = ts.TwoAdicFriPcs(ts.FriConfig(2, 28, 8), ctx)
trace = generate_synth_mul_trace(1 << 20)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    root, data = pcs.commit([((20, 1), trace)])
    del data
ctx.synchronize()
print("ok", root[:2])
