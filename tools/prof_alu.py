"""Runs the library's pure-ALU loops (ts_bench_alu: butterflies, Blake3, SHA-256) a few times; under
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace this gives the clock and
the VALU issue rate the chip sustains with no memory traffic, to set beside the prover's kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tapstark_amd as ts
ctx = ts.default_context()
for kind in (0, 1, 2):
    for _ in range(3):
        r = ctx.alu_ceiling(kind)
    print(kind, r)
