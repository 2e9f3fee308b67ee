"""Time of BFMmcs::commit on resident matrices (leaf hashes + every tree level), per shape, from the
library's per-kernel HIP events.  Knobs come from the environment (TS_LEAF_TREE, TS_LEAF_TREE_R,
TS_LEAF_TREE_FINISH, TS_TREE_MAX_LOG).

    python tools/time_leaf_tree.py [log_leaves ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd.airs import splitmix64_stream  # noqa: E402

ctx = ts.default_context()
mm = ts.Blake3Mmcs(ctx)
tag = {k: v for k, v in os.environ.items() if k.startswith("TS_LEAF_TREE") or k == "TS_TREE_MAX_LOG"}
logs = [int(a) for a in sys.argv[1:]] or [18, 20, 22]
for lg in logs:
    for label, widths in (("w=2", [2]), ("w=4+4", [4, 4]), ("w=64", [64])):
        mats = [splitmix64_stream(1 + i, (1 << lg) * w).reshape(1 << lg, w) for i, w in enumerate(widths)]
        root0 = mm.commit([m.copy() for m in mats])[0]
        ctx.set_kernel_timing(True)
        reps = 5
        for _ in range(reps):
            mm.commit([m.copy() for m in mats])
        kt = ctx.take_kernel_timings()
        ctx.set_kernel_timing(False)
        ks = {k: v for k, v in kt.items() if "leaf" in k or "merkle" in k}
        tot = sum(v[1] for v in ks.values()) / reps
        comp = (1 << lg) * (sum(widths) + 15) // 16 + (1 << lg)
        print(f"{tag} 2^{lg} {label:6s}: {1e3 * tot:8.1f} us  {comp / tot / 1e6:6.1f} G compressions/s  "
              f"{ {k.replace('k_', ''): (v[0] // reps, round(1e3 * v[1] / reps, 1)) for k, v in ks.items()} }  "
              f"root {np.asarray(root0).ravel()[:2]}", flush=True)
