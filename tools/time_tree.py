"""Per-tree time of the Merkle kernels above the leaf level, 2^12..2^22 leaves, and of whole FRI commit
rounds inside a proof.  TS_TREE_MAX_LOG=16 selects per-level launches down to 2^16 nodes (round-2 shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
from tapstark_amd.airs import splitmix64_stream

ctx = ts.default_context()
mm = ts.Blake3Mmcs(ctx)
tag = os.environ.get("TS_TREE_MAX_LOG", "22")
for lg in (12, 14, 16, 17, 18, 19, 20, 21, 22):
    m = splitmix64_stream(1, (1 << lg) * 2).reshape(1 << lg, 2)
    dm = ts.DeviceMatrix.from_host(ctx, m) if hasattr(ts.DeviceMatrix, "from_host") else None
    mm.commit([m.copy()])
    ctx.set_kernel_timing(True)
    reps = 10
    for _ in range(reps):
        mm.commit([m.copy()])
    kt = ctx.take_kernel_timings()
    ctx.set_kernel_timing(False)
    tot = sum(v[1] for k, v in kt.items() if "merkle" in k) / reps
    print(f"max_log={tag} 2^{lg}: merkle kernels {1e3 * tot:8.1f} us/tree ",
          {k.split("(")[0][-20:]: (v[0] // reps, round(1e3 * v[1] / reps, 1)) for k, v in kt.items() if "merkle" in k})
