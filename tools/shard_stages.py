"""Where the time of ONE sharded proof goes, measured on a one-GPU box (BASELINE configs 4 / 5 over G ranks).

The G ranks are threads of this process on the box's one GPU (ts_comm_local_*).  Running them
concurrently says little (they share the card), so the ranks take TURNS: a token is held by a rank
whenever it is outside a collective -- it synchronises its stream and hands the token on when it
enters one, and takes it back when the collective returns.  A rank's kernels therefore run alone on
the GPU, as they would on its own GPU, and for every segment between two collectives the host time
each rank held the token is its solo time for that segment.  From these:

  critical_path_ms   = sum over segments of the slowest rank's time   (collectives are barriers)
  per-rank kernel ms = HIP-event sums per kernel (one more proof with the kernel timers on)
  collectives        = kind / bytes / count of every collective of the proof (stage timers)

What a one-GPU box cannot give is the cost of the collectives themselves over xGMI: `model` prices
them with stated assumptions (latency per small collective, link bandwidth for the bulk ones).

    python tools/shard_stages.py config4 [G] [variant ...] > profiles/r04_config4_shard_stages.json
    variants: replicated (min_local_log 12), localq, mll16, mll20, localq16 (localq + mll16)
"""
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd import _lib  # noqa: E402
from tapstark_amd.airs import SynthExtAir, SynthMulAir  # noqa: E402
from tapstark_amd.benchutil import split_stage_timings  # noqa: E402
from tapstark_amd.comm import LocalCommGroup  # noqa: E402

CONFIGS = {"config4": ("mul64", 22, 64, (4, 16, 8)), "config5": ("ext163", 20, 163, (4, 16, 8)),
           "config3": ("mul64", 20, 64, (2, 28, 8))}


class TurnComm:
    """Wraps a native in-process communicator: the rank holds `token` except inside collectives."""

    def __init__(self, inner, ctx, token):
        self.inner, self.rank, self.world, self.error = inner, inner.rank, inner.world, None
        self.segments = []       # host seconds the token was held, one entry per segment
        self._t = None
        ic = inner.c

        def leave():
            ctx.synchronize()    # the rank's GPU work of this segment is done
            self.segments.append(time.perf_counter() - self._t)
            token.release()

        def enter():
            token.acquire()
            self._t = time.perf_counter()

        def ag(_user, send, recv, nbytes, stream):
            leave()
            try:
                return ic.all_gather(ic.user, send, recv, nbytes, stream)
            finally:
                enter()

        def bc(_user, buf, nbytes, root, stream):
            leave()
            try:
                return ic.broadcast(ic.user, buf, nbytes, root, stream)
            finally:
                enter()

        self._ag, self._bc = _lib.ALL_GATHER_FN(ag), _lib.BROADCAST_FN(bc)
        self.c = _lib.CommC(ic.rank, ic.world, ic.user, self._ag, self._bc, ic.abort)
        self.enter, self.leave = enter, leave


def run(name, G, variant):
    kind, log_n, w, cfg = CONFIGS[name]
    n = 1 << log_n
    air = SynthMulAir(64) if kind == "mul64" else SynthExtAir(163)
    tape = ts.air_tape(air, 0)
    group = LocalCommGroup(G)
    token = threading.Lock()
    kw = dict(trace_replicated=True,
              local_quotient=variant.startswith("localq"),
              min_local_log={"mll16": 16, "mll20": 20, "localq16": 16}.get(variant, 12))
    out, errs = [None] * G, [None] * G

    def gen(ctx):
        return ts.DeviceMatrix.synth_mul(ctx, n, w) if kind == "mul64" else ts.DeviceMatrix.synth_ext(ctx, n, w)

    def rank_main(r):
        try:
            ctx = ts.Context(0)
            config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
            cair = ts.CompiledAir(ctx, tape)
            comm = TurnComm(group.comm(r), ctx, token)
            res = {}
            # proof 0 builds tables and grows the pool; 1-3: segments (no timers); then stage timers
            # (collectives table); then kernel timers
            for mode in ("warm", "segments", "segments", "segments", "stages", "kernels"):
                m = gen(ctx)
                ctx.synchronize()
                ctx.set_timing(mode == "stages")
                ctx.set_kernel_timing(mode == "kernels")
                comm.segments = []
                comm.enter()
                p = ts.prove_sharded(config, cair, ts.BfChallenger(), m, [], comm, **kw)
                comm.leave()
                if mode == "segments":  # three passes, element-wise minimum: host hiccups drop out
                    cur = [round(1e3 * s, 4) for s in comm.segments]
                    prev = res.get("segments_ms")
                    res["segments_ms"] = cur if prev is None or len(prev) != len(cur) else [min(a, b) for a, b in zip(prev, cur)]
                if mode == "stages":
                    res["stages_ms"], res["collectives"] = split_stage_timings(ctx.take_timings())
                    ctx.set_timing(False)
                if mode == "kernels":
                    kt = ctx.take_kernel_timings()
                    ctx.set_kernel_timing(False)
                    res["kernels_ms"] = {k: round(v[1], 4) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])}
                    res["kernel_ms_total"] = round(sum(v[1] for v in kt.values()), 3)
            res["proof_sha256"] = __import__("hashlib").sha256(p.words.tobytes()).hexdigest()
            out[r] = res
        except BaseException as e:  # noqa: BLE001
            errs[r] = repr(e)
            try:
                token.release()
            except RuntimeError:
                pass

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not any(errs), errs
    return out, (n, w, cfg)


def model(colls, G):
    """Prices the collectives of one proof over xGMI with explicit assumptions (NOT measured here):
    a small collective (< 64 KiB) costs `lat_us`; a bulk all-gather moves bytes_per_rank to each of
    the G-1 peers over its own link at `link_GBps` (direct schedule, links in parallel); a broadcast
    of B bytes from one rank leaves through its G-1 links in parallel, B per link."""
    lat_us, link = 25.0, 120.0  # RCCL small-message latency on MI300-class nodes; ~80 % of 153 GB/s per link
    total, rows = 0.0, []
    for c in colls:
        small = c["bytes"] < (64 << 10)
        per = lat_us * 1e-3 if small else lat_us * 1e-3 + c["bytes"] / (link * 1e9) * 1e3
        rows.append({"what": c["what"], "count": c["count"], "est_ms_each": round(per, 4),
                     "est_ms_total": round(per * c["count"], 4)})
        total += per * c["count"]
    return {"assumptions": {"small_collective_latency_us": lat_us, "link_GB_per_s": link,
                            "schedule": "direct: every pair of ranks has its own xGMI link"},
            "rows": rows, "collectives_ms_total": round(total, 3)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config4"
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    variants = sys.argv[3:] or ["replicated", "localq", "mll16", "localq16", "mll20"]
    res = {"_comment": __doc__.split("\n\n")[0] + " " + " ".join(__doc__.split("\n\n")[1].split()),
           "config": name, "ranks": G}
    shas = set()
    for v in variants:
        ranks, (n, w, cfg) = run(name, G, v)
        segs = [r["segments_ms"] for r in ranks]
        n_seg = min(len(s) for s in segs)
        assert all(len(s) == n_seg for s in segs), "ranks disagree on the number of collectives"
        crit = [max(s[k] for s in segs) for k in range(n_seg)]
        mdl = model(ranks[0]["collectives"], G)
        shas.update(r["proof_sha256"] for r in ranks)
        # the segments that dominate, named by the collective that ends them
        order = []
        for c in ranks[0]["collectives"]:
            order.append(c["what"])
        res[v] = {
            "shape": {"rows": n, "width": w, "log_blowup": cfg[0], "queries": cfg[1]},
            "segments": n_seg, "collectives_per_proof": n_seg - 1,
            "critical_path_compute_ms": round(sum(crit), 3),
            "per_rank_compute_ms": [round(sum(s), 3) for s in segs],
            "slowest_rank_per_segment_ms": [round(x, 3) for x in crit],
            "per_rank_kernel_ms_total": [r["kernel_ms_total"] for r in ranks],
            "collectives_rank0": ranks[0]["collectives"],
            "xgmi_model": mdl,
            "estimate_ms_per_proof_on_G_gpus": round(sum(crit) + mdl["collectives_ms_total"], 2),
            "rank0_kernels_ms": ranks[0]["kernels_ms"], "rank1_kernels_ms": ranks[1 % G]["kernels_ms"],
            "rank0_stages_ms": ranks[0]["stages_ms"],
        }
    res["all_variants_same_proof"] = len(shas) == 1
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
