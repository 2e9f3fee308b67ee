"""Soak of the sharded prover on the in-process communicator: G rank threads prove the same trace
K times (row-sliced and replicated inputs, both inverse-NTT options and both quotient paths in turn); every proof of every
rank must equal ts_prove's.  Exercises the rendezvous of csrc/comm.cpp (generation counting, buffer
reuse across collectives) far beyond what the test suite does.

    python tools/soak_sharded.py [K] [G] [log_n]
"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd.airs import SynthMulAir, generate_synth_mul_trace  # noqa: E402
from tapstark_amd.comm import LocalCommGroup  # noqa: E402


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    log_n = int(sys.argv[3]) if len(sys.argv) > 3 else 13
    n, w, cfg = 1 << log_n, 24, (3 if G <= 8 else 4, 9, 4)
    air = SynthMulAir(w)
    trace = generate_synth_mul_trace(n, w, 5)
    ctx0 = ts.default_context()
    ref = ts.prove(ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx0)), air, ts.BfChallenger(),
                   trace.copy(), []).words.tobytes()
    group = LocalCommGroup(G)
    bad, errs = [], []

    def rank_main(r):
        try:
            ctx = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
            cair = ts.CompiledAir(ctx, ts.air_tape(air, 0))
            rows = np.ascontiguousarray(trace[r * n // G:(r + 1) * n // G])
            for k in range(K):
                repl = bool(k & 1)
                m = ts.DeviceMatrix.upload(ctx, trace if repl else rows)
                p = ts.prove_sharded(config, cair, ts.BfChallenger(), m, [], group.comm(r),
                                     trace_replicated=repl,
                                     local_quotient=bool(k & 4))
                if p.words.tobytes() != ref:
                    bad.append((r, k))
        except BaseException as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    print(f"soak_sharded: {K} proofs x {G} ranks (2^{log_n} x {w}, log_blowup {cfg[0]}), mismatches: {len(bad)} {bad[:5]}, errors: {errs[:3]}")
    sys.exit(1 if bad or errs else 0)


if __name__ == "__main__":
    main()
