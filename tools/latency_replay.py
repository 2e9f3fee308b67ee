"""The ceiling of a "zero-host-sync" proof, measured: single-proof latency with the
host's four mid-proof synchronisations as they are, against the same proof with those synchronisations
skipped and their results handed to the host at once (ts_ctx_set_replay: recorded from an identical proof,
so the challenges cost nothing at all -- no device-side sponge, no extra launches).  Settings alternate
proof by proof on one box; every proof must equal the recorded one.

    python tools/latency_replay.py [reps=20]  ->  stdout
"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

import bench  # noqa: E402
import tapstark_amd as ts  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = ts.default_context()
for name in ("config3", "config2"):
    air, _, pis, desc, cfg, shape, gen = bench.workload(name, 20, False)
    if callable(pis):
        pis = np.array([0, 1, pis(ctx)], dtype=np.uint32)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    for _ in range(3):
        ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis)
    ctx.set_replay(1)
    ref = ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis).words.copy()
    t = {0: [], 2: []}
    for i in range(reps):
        for mode in ((0, 2) if i % 2 == 0 else (2, 0)):
            m = gen(ctx)
            ctx.synchronize()
            ctx.set_replay(mode)
            t0 = time.perf_counter()
            p = ts.prove(config, cair, ts.BfChallenger(), m, pis)
            t[mode].append(1e3 * (time.perf_counter() - t0))
            assert len(p.words) == len(ref) and (p.words == ref).all(), f"mode {mode}: proof differs"
    ctx.set_replay(0)
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    lo = {k: min(v) for k, v in t.items()}
    print(f"{desc}\n  single-proof latency, {reps} proofs each, alternating, all bit-identical:\n"
          f"    host synchronises 4x mid-proof (as shipped)   median {med[0]:.3f} ms   min {lo[0]:.3f}\n"
          f"    no mid-proof synchronisation (results replayed) median {med[2]:.3f} ms   min {lo[2]:.3f}\n"
          f"    -> ceiling of a zero-host-sync proof: {med[0] - med[2]:+.3f} ms ({100 * (med[0] - med[2]) / med[0]:.1f} % of latency), "
          "before the cost of sampling the challenges on the device", flush=True)
