"""Single-proof latency (host clock around ts_prove, median of N, no timers inside) and the per-proof
kernel-time sum, for config 3 and config 2.  Knobs under test come from the environment
(TS_TREE_MAX_LOG, TS_FRI_ROUND_LOG)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
import bench

tag = {k: os.environ[k] for k in ("TS_TREE_MAX_LOG", "TS_FRI_ROUND_LOG", "TS_FRI_GRAPH") if k in os.environ}
ctx = ts.default_context()
for name in sys.argv[1:] or ("config3", "config2"):
    air, _, pis, desc, cfg, shape, gen = bench.workload(name, 20, False)
    if callable(pis):
        pis = np.array([0, 1, pis(ctx)], dtype=np.uint32)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    ref = ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis).words.tobytes()
    for _ in range(3):
        ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis)
    lat = []
    for _ in range(15):
        m = gen(ctx)
        ctx.synchronize()
        t0 = time.perf_counter()
        p = ts.prove(config, cair, ts.BfChallenger(), m, pis)
        lat.append(1e3 * (time.perf_counter() - t0))
    assert p.words.tobytes() == ref
    ctx.set_kernel_timing(True)
    for _ in range(3):
        ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis)
    kt = ctx.take_kernel_timings()
    ctx.set_kernel_timing(False)
    ksum = sum(v[1] for v in kt.values()) / 3
    nk = sum(v[0] for v in kt.values()) // 3
    fri = sum(v[1] for k, v in kt.items() if "fri" in k or "merkle" in k) / 3
    import hashlib
    print(f"{name} {tag}: latency median {sorted(lat)[len(lat)//2]:.3f} ms  min {min(lat):.3f}  "
          f"kernels {ksum:.3f} ms in {nk} launches  (fri+merkle kernels {fri:.3f})  proof sha {hashlib.sha256(ref).hexdigest()[:12]}"
          f"  graph {ctx.graph_stats()}")
