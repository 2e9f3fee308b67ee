"""Times the taptree MMCS kernels (run under rocprofv3 --kernel-trace --stats for profiles/):
    python tools/prof_taptree.py [log_h width u32 Q]
prints a JSON line with the SHA-256 compression rate of the leaf kernel."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd import taptree as tt  # noqa: E402
from tapstark_amd.airs import splitmix64_stream  # noqa: E402


def main():
    log_h, width, u32, Q = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (16, 8, 4, 16)))
    ctx = ts.default_context()
    n_evals = width // u32
    lock_cache = {}

    def locks(q, n):
        out = []
        for s in range(1 + n):
            key = (q, s)
            if key not in lock_cache:
                lock_cache[key] = tt.winternitz_lock_script(bytes([q & 255, s & 255, s >> 8]), 1 if s == 0 else u32)
            out.append(lock_cache[key])
        return out

    m = splitmix64_stream(5, (1 << log_h) * width).reshape(1 << log_h, width)
    mm = tt.TapTreeMmcs(Q, locks, u32_size=u32, ctx=ctx)
    mm.commit([m.copy()])  # warm-up (lock scripts, pool)
    ctx.set_kernel_timing(True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        roots, data = mm.commit([ts.DeviceMatrix.upload(ctx, m)])
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps
    kt = ctx.take_kernel_timings()
    script_len = sum(len(x) for x in locks(0, n_evals)) + 5 + n_evals * u32 * 6 + 1
    compress_per_leaf = (64 + 4 + script_len + 9 + 63) // 64
    # with the prefix table (default; TS_TAP_PREFIX=0 switches it off) a leaf starts after the index
    # lock script: those blocks are looked up, not compressed
    prefix = os.environ.get("TS_TAP_PREFIX", "1") != "0"
    executed = compress_per_leaf - ((4 + len(locks(0, n_evals)[0])) // 64 if prefix else 0)
    leaves = (1 << log_h) * Q
    leaf_ms = kt["k_tapleaf_template"][1] / kt["k_tapleaf_template"][0]
    branch_ms = kt["k_tapbranch_level"][1] / reps
    print(json.dumps({
        "shape": {"log_height": log_h, "width": width, "u32_size": u32, "num_queries": Q},
        "leaf_script_bytes": script_len, "compressions_per_leaf": compress_per_leaf,
        "k_tapleaf_template_ms": round(leaf_ms, 4),
        "prefix_table": prefix, "compressions_executed_per_leaf": executed,
        "sha256_compressions_per_s": round(leaves * executed / (leaf_ms * 1e-3), 0),
        "sha256_compressions_per_s_alu_peak": round(ctx.alu_ceiling(2), 0),
        "leaf_script_GB_per_s": round(leaves * script_len / (leaf_ms * 1e-3) / 1e9, 2),
        "k_tapbranch_levels_ms": round(branch_ms, 4),
        "commit_wall_ms": round(wall * 1e3, 3),
    }))


if __name__ == "__main__":
    main()
