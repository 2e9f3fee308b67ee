"""Soak: S lanes proving concurrently for a while; every proof of a trace must equal the first proof of
that trace (the prover is deterministic) and must verify.  soak.py [proofs_per_lane] [lanes]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
from tapstark_amd.airs import SynthMulAir

per_lane = int(sys.argv[1]) if len(sys.argv) > 1 else 30
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, w, cfg = 1 << 20, 64, (2, 28, 8)
air = SynthMulAir(w)
tape = ts.air_tape(air, 0)
lanes = []
for _ in range(S):
    c = ts.Context(0)
    lanes.append((c, ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c)), ts.CompiledAir(c, tape)))
seeds = [0x7A957A12, 12345, 99]
ref = {}
for sd in seeds:
    c, conf, ca = lanes[0]
    ref[sd] = ts.prove(conf, ca, ts.BfChallenger(), ts.DeviceMatrix.synth_mul(c, n, w, sd), []).words
    ts.verify(conf, ca, ts.BfChallenger(), ref[sd], [])
bad = []
def job(l):
    c, conf, ca = lanes[l]
    for i in range(per_lane):
        sd = seeds[(i + l) % len(seeds)]
        p = ts.prove(conf, ca, ts.BfChallenger(), ts.DeviceMatrix.synth_mul(c, n, w, sd), [])
        if len(p.words) != len(ref[sd]) or not (p.words == ref[sd]).all():
            bad.append((l, i, sd))
ths = [threading.Thread(target=job, args=(l,)) for l in range(S)]
[t.start() for t in ths]; [t.join() for t in ths]
print(f"soak: {S * per_lane} proofs on {S} lanes, mismatches: {len(bad)} {bad[:5]}")
sys.exit(1 if bad else 0)
