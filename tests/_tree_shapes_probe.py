"""Helper of test_gpu_mmcs.py::test_tree_launch_shapes_agree (run as a child process, because the
launch-shape knobs are read once per process): prints the roots of a few trees, an opened path and
the hash of a proof as JSON."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import tapstark_amd as ts
from tapstark_amd.airs import FibonacciAir, SynthMulAir, generate_synth_mul_trace, splitmix64_stream


def probe():
    ctx = ts.default_context()
    mm = ts.Blake3Mmcs(ctx)
    out = {}
    for lg in (0, 3, 9, 13, 17, 18, 19, 20, 22):
        m = splitmix64_stream(100 + lg, (1 << lg) * 2).reshape(1 << lg, 2)
        root, data = mm.commit([m])
        rows, path = mm.open_batch((1 << lg) // 3, data)
        out[f"tree{lg}"] = [int(x) for x in root] + [int(x) for x in np.asarray(path).ravel()]
    for lg, widths in ((11, [4, 4]), (19, [64]), (20, [4, 4, 4, 4]), (21, [17])):  # table / strided leaves, ragged block
        mats = [splitmix64_stream(7 * lg + i, (1 << lg) * w).reshape(1 << lg, w) for i, w in enumerate(widths)]
        root, data = mm.commit(mats)
        rows, path = mm.open_batch((1 << lg) - 5, data)
        out[f"wide{lg}"] = [int(x) for x in root] + [int(x) for x in np.asarray(path).ravel()]
    n = 1 << 19  # FRI rounds with 2^20 .. 2^10 leaves at log_blowup 2
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 9, 4), ctx))
    last = int(ts.DeviceMatrix.fibonacci(ctx, 0, 1, n).download()[-1, 1])
    pis = np.array([0, 1, last], dtype=np.uint32)
    cair = ts.CompiledAir(ctx, ts.air_tape(FibonacciAir(), 3))
    proof = ts.prove(config, cair, ts.BfChallenger(), ts.DeviceMatrix.fibonacci(ctx, 0, 1, n), pis)
    ts.verify(config, cair, ts.BfChallenger(), proof, pis)
    out["proof"] = hashlib.sha256(proof.words.tobytes()).hexdigest()
    # a degree-3 AIR: two quotient chunks, whose LDEs share one set of launches (TS_LDE_PAIR)
    air = SynthMulAir(3)
    cair = ts.CompiledAir(ctx, ts.air_tape(air, 0))
    trace = generate_synth_mul_trace(1 << 13, 3)
    proof = ts.prove(config, cair, ts.BfChallenger(), trace, np.zeros(0, dtype=np.uint32))
    ts.verify(config, cair, ts.BfChallenger(), proof, np.zeros(0, dtype=np.uint32))
    out["proof_mul3"] = hashlib.sha256(proof.words.tobytes()).hexdigest()
    return out


if __name__ == "__main__":
    print("PROBE " + json.dumps(probe()))
