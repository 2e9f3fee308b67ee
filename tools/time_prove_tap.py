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
