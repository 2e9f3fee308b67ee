import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tapstark_amd as ts
from tapstark_amd.airs import SynthMulAir
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, w, cfg = 1 << 20, 64, (2, 28, 8)
tape = ts.air_tape(SynthMulAir(w), 0)
lanes = []
for _ in range(S):
    c = ts.Context(0)
    lanes.append((c, ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c)), ts.CompiledAir(c, tape)))
host = ts.DeviceMatrix.synth_mul(lanes[0][0], n, w).download()
pins = []
for _ in range(S):
    p = ts.PinnedHostMatrix(n, w); p.array[:] = host; pins.append(p)
lock = threading.Lock(); last = [-1e9]; gap = [0.0]
def gate():
    if gap[0] <= 0: return
    with lock:
        while True:
            wt = last[0] + gap[0] - time.perf_counter()
            if wt <= 0: break
            time.sleep(min(wt, 2e-4))
        last[0] = time.perf_counter()
def one(l):
    c, conf, ca = lanes[l]
    m = ts.DeviceMatrix.upload_async(c, pins[l])
    gate()
    return ts.prove(conf, ca, ts.BfChallenger(), m, [])
for l in range(S): one(l)
t0 = time.perf_counter(); one(0); solo = time.perf_counter() - t0
gap[0] = 0.25 * solo
def job(l):
    for _ in range(per): one(l)
for rep in range(2):
    ths = [threading.Thread(target=job, args=(l,)) for l in range(S)]
    t0 = time.perf_counter()
    [t.start() for t in ths]; [t.join() for t in ths]
    for c, _, _ in lanes: c.synchronize()
    dt = time.perf_counter() - t0
    print(f"python pinned stream, {S} lanes x {per}: {1e3*dt/(S*per):.3f} ms per proof ({n*w*4*S*per/dt/1e9:.1f} GB/s), gate {1e3*gap[0]:.2f} ms")
