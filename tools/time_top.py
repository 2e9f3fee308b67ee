"""Times the Merkle-top kernel on a 2^16-leaf tree (TS_TOP_VAR selects timing variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
from tapstark_amd.airs import splitmix64_stream
ctx = ts.default_context()
mm = ts.Blake3Mmcs(ctx)
for lg in (16, 8, 12):
    m = splitmix64_stream(1, (1 << lg) * 2).reshape(1 << lg, 2)
    mm.commit([m.copy()])
    ctx.set_kernel_timing(True)
    for _ in range(20):
        mm.commit([m.copy()])
    kt = ctx.take_kernel_timings()
    ctx.set_kernel_timing(False)
    print(lg, os.environ.get("TS_TOP_VAR", "0"), {k: round(1e3 * v[1] / v[0], 2) for k, v in kt.items() if "merkle" in k or "leaf" in k})
