"""Single-proof latency (host clock around ts_prove, no timers inside) for config 3 and config 2, with
the FRI commit phase eager and replayed from a hipGraph (TS_FRI_GRAPH is read per proof, so the two
settings ALTERNATE proof by proof inside one process, on one context: same box, same clocks, same
pool), plus the per-proof kernel-time sum.  Other knobs under test come from the environment
(TS_TREE_MAX_LOG, TS_FRI_ROUND_LOG).

    python tools/latency.py [config3 config2]
"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import tapstark_amd as ts  # noqa: E402

tag = {k: os.environ[k] for k in ("TS_TREE_MAX_LOG", "TS_FRI_ROUND_LOG") if k in os.environ}
os.environ.pop("TS_FRI_GRAPH", None)
ctx = ts.default_context()
N = 40


def med(xs):
    return sorted(xs)[len(xs) // 2]


for name in sys.argv[1:] or ("config3", "config2"):
    air, _, pis, desc, cfg, shape, gen = bench.workload(name, 20, False)
    if callable(pis):
        pis = np.array([0, 1, pis(ctx)], dtype=np.uint32)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    ref = ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis).words.tobytes()
    lat = {"eager": [], "graph": []}
    for i in range(-6, 2 * N):  # the first six prime both settings (recording proof, pool, graph instantiation)
        mode = "graph" if i % 2 else "eager"
        if mode == "graph":
            os.environ["TS_FRI_GRAPH"] = "1"
        else:
            os.environ.pop("TS_FRI_GRAPH", None)
        m = gen(ctx)
        ctx.synchronize()
        t0 = time.perf_counter()
        p = ts.prove(config, cair, ts.BfChallenger(), m, pis)
        dt = 1e3 * (time.perf_counter() - t0)
        assert p.words.tobytes() == ref, f"{mode}: proof differs"
        if i >= 0:
            lat[mode].append(dt)
    os.environ.pop("TS_FRI_GRAPH", None)
    ctx.set_kernel_timing(True)
    for _ in range(3):
        ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis)
    kt = ctx.take_kernel_timings()
    ctx.set_kernel_timing(False)
    ksum = sum(v[1] for v in kt.values()) / 3
    nk = sum(v[0] for v in kt.values()) // 3
    fri = sum(v[1] for k, v in kt.items() if "fri" in k or "merkle" in k) / 3
    e, g = lat["eager"], lat["graph"]
    print(f"{name} {tag}: {N} proofs per setting, alternating.  eager: median {med(e):.3f} ms, min {min(e):.3f};  "
          f"TS_FRI_GRAPH=1: median {med(g):.3f} ms, min {min(g):.3f};  graph - eager (medians) {med(g) - med(e):+.3f} ms.  "
          f"kernels {ksum:.3f} ms in {nk} launches (fri + merkle kernels {fri:.3f});  proof sha "
          f"{hashlib.sha256(ref).hexdigest()[:12]};  graph stats {ctx.graph_stats()}")
