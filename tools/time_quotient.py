import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bench, tapstark_amd as ts
ctx = ts.default_context()
for name in ("config3", "config5", "config2"):
    air, _, pis, desc, cfg, shape, gen = bench.workload(name, 20, False)
    if callable(pis): pis = np.array([0, 1, pis(ctx)], dtype=np.uint32)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis)
    ctx.set_kernel_timing(True)
    for _ in range(8): ts.prove(config, cair, ts.BfChallenger(), gen(ctx), pis)
    kt = ctx.take_kernel_timings(); ctx.set_kernel_timing(False)
    print(os.environ.get("TS_JIT_NO_TRACK", "track"), name, {k: round(v[1] / 8 * 1e3, 1) for k, v in kt.items() if "quotient" in k})
