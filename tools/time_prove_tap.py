"""Times prove() over taptrees (ts_prove_tap) on a Fibonacci trace: time_prove_tap.py [log_n] [Q] [log_blowup]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
from tapstark_amd import taptree as tt
from tapstark_amd.airs import FibonacciAir

log_n, Q, b = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (16, 28, 2)))
ctx = ts.default_context()
n = 1 << log_n
cache = {}
def lock_for(ci, q, s, u32):
    k = (ci, q, s, u32)
    if k not in cache:
        cache[k] = tt.winternitz_lock_script(bytes([ci & 255, q & 255, s & 255, ci >> 8]), u32)
    return cache[k]
locks = tt.make_lock_table(Q, 2, 1, log_n, lock_for)
config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(b, Q, 8), ctx))
air = FibonacciAir()
last = int(ts.DeviceMatrix.fibonacci(ctx, 0, 1, n).download()[-1, 1])
pis = np.array([0, 1, last], dtype=np.uint32)
cair = ts.CompiledAir(ctx, ts.air_tape(air, 3))
tt.prove_tap(config, cair, ts.BfChallenger(), ts.DeviceMatrix.fibonacci(ctx, 0, 1, n), pis, locks)  # warm-up
ctx.set_kernel_timing(True)
t0 = time.perf_counter()
proof = tt.prove_tap(config, cair, ts.BfChallenger(), ts.DeviceMatrix.fibonacci(ctx, 0, 1, n), pis, locks)
dt = time.perf_counter() - t0
kt = ctx.take_kernel_timings()
ok = tt.verify_tap(config, air, ts.BfChallenger(), proof, pis, locks)
top = sorted(kt.items(), key=lambda kv: -kv[1][1])[:4]
print(json.dumps({"workload": f"Fibonacci 2^{log_n}x2, log_blowup {b}, {Q} queries, TapTreeMmcs", "prove_ms": round(dt * 1e3, 2),
                  "verdict": ok, "proof_words": int(len(proof)), "lock_table_MB": round(sum(map(len, locks)) / 1e6, 2),
                  "top_kernels_ms": {k: round(v[1], 2) for k, v in top}}))

# optional 4th argument G: the same proof with its trees split over G rank threads on this one GPU
# (ts_prove_tap_sharded on the in-process communicator).  The ranks share the card, so wall time says
# nothing; what the line shows is how the SHA-256 work -- a rank's leaf-kernel time -- divides.
if len(sys.argv) >= 5:
    import threading
    from tapstark_amd.comm import LocalCommGroup
    G = int(sys.argv[4])
    group = LocalCommGroup(G)
    res = [None] * G
    host_trace = ts.DeviceMatrix.fibonacci(ctx, 0, 1, n).download()

    def rank_main(r):
        c = ts.Context(0)
        conf = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(b, Q, 8), c))
        ca = ts.CompiledAir(c, ts.air_tape(air, 3))
        for timed in (False, True):
            c.set_kernel_timing(timed)
            p = tt.prove_tap(conf, ca, ts.BfChallenger(), host_trace.copy(), pis, locks, comm=group.comm(r))
        k = c.take_kernel_timings()
        res[r] = (bool((p == proof).all()), round(k.get("k_tapleaf_template", (0, 0.0))[1], 2),
                  round(sum(v[1] for v in k.values()), 2))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    per = -(-Q // G)
    print(json.dumps({"split_by_tree_over_ranks": G, "trees_per_rank": [max(0, min(Q, (r + 1) * per) - r * per) for r in range(G)],
                      "proof_identical": [x[0] for x in res], "leaf_kernel_ms_per_rank": [x[1] for x in res],
                      "all_kernels_ms_per_rank": [x[2] for x in res],
                      "note": "G threads on ONE GPU: kernel times are inflated by sharing the card; the split is the point"}))
