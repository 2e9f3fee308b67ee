"""Single-proof latency (wall clock around ts_prove, one proof alone on the GPU) of configs 3 and 2: median / min /
p90 / max of N proofs.  Settings that are read once per process (TS_SYNC_SPIN ...) are compared by running this
in separate processes, alternating (tools/ab_sync_spin.sh).   python tools/latency_simple.py [n=40]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

import bench  # noqa: E402
import tapstark_amd as ts  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = ts.default_context()
out = []
for name in ("config3", "config2"):
    air, _, pis, desc, cfg, shape, gen = bench.workload(name, 20, False)
    if callable(pis):
        pis = np.array([0, 1, pis(ctx)], dtype=np.uint32)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    for _ in range(5):
        ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis)
    lat = []
    for _ in range(n):
        m = gen(ctx)
        ctx.synchronize()
        t0 = time.perf_counter()
        ts.prove(config, cair, ts.BfChallenger(), m, pis)
        lat.append(1e3 * (time.perf_counter() - t0))
    lat.sort()
    out.append(f"{name}: median {lat[n // 2]:.3f} min {lat[0]:.3f} p90 {lat[int(0.9 * n)]:.3f} max {lat[-1]:.3f}")
print(f"TS_SYNC_SPIN={os.environ.get('TS_SYNC_SPIN', '0')}  " + "   ".join(out), flush=True)
