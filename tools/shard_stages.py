"""Per-rank stage timings of ONE proof sharded over G ranks (BASELINE config 4's split), with the
ranks as G threads of this process on the one GPU of the box (ts_comm_local_*).  The ranks share the
card, so absolute times are ~G x what a rank alone on its GPU would take and the collectives are
device-to-device copies, not xGMI; what the table shows is the SPLIT of a rank's GPU time over the
stages, for both ways of doing the per-column part of the inverse NTT.  A stage that ends in a
collective (Merkle commits, quotient broadcast, FRI rounds, the query phase's all-gather of answers)
includes the wait for the slowest rank, which on a shared card is scheduling skew: e.g. a query
phase of 8 ms on some ranks and 0.5 ms on the others is 0.5 ms of work.

    python tools/shard_stages.py [log_n] [G] > profiles/r02_config4_shard_stages.json
"""
import json
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd.airs import SynthMulAir  # noqa: E402
from tapstark_amd.comm import LocalCommGroup  # noqa: E402


def run(log_n, G, colshard, cfg=(4, 16, 8)):
    n = 1 << log_n
    air = SynthMulAir(64)
    group = LocalCommGroup(G)
    out, errs = [None] * G, [None] * G

    def rank_main(r):
        try:
            ctx = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
            cair = ts.CompiledAir(ctx, ts.air_tape(air, 0))
            # the first proof builds tables and grows the pool, the second does the same with the
            # stage timers on (they synchronise at stage boundaries); the third is the one reported
            for timed in (False, True, True):
                ctx.set_timing(timed)
                ctx.take_timings()
                p = ts.prove_sharded(config, cair, ts.BfChallenger(), ts.DeviceMatrix.synth_mul(ctx, n, 64), [],
                                     group.comm(r), trace_replicated=True, column_sharded_inverse=colshard)
            st = {}
            for k, v in ctx.take_timings():
                st[k] = round(st.get(k, 0.0) + v, 3)
            out[r] = (st, int(p.words[-1]))
        except BaseException as e:  # noqa: BLE001
            errs[r] = repr(e)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not any(errs), errs
    return [o[0] for o in out]


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    res = {"_comment": __doc__.split("\n\n")[0], "log_n": log_n, "width": 64, "log_blowup": 4, "ranks": G}
    for name, cs in (("replicated_inverse", False), ("column_sharded_inverse", True)):
        stages = run(log_n, G, cs)
        res[name] = {"per_rank_ms": stages}
        r0 = stages[0]
        total = r0.get("prove", 0.0)
        rep = sum(v for k, v in r0.items() if "every column on every rank" in k or
                  (k == "lde: inverse NTT, contiguous stages" and not cs))
        res[name]["rank0_replicated_ms"] = round(rep, 3)
        res[name]["rank0_replicated_frac_of_prove"] = round(rep / total, 4) if total else None
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
