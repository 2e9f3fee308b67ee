#!/usr/bin/env python3
"""bench.py -- trace cells/sec (and proofs/sec) of the MI355X STARK prover hot path.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete prove() (commit trace -> quotient -> commit chunks -> open -> FRI ->
queries -> proof on the host) of BASELINE.json configs[2]: the build-defined SynthMulAir-64 trace,
2^20 rows x 64 columns, log_blowup 2, 28 queries, 8 PoW bits, with the trace already resident in
HBM when the timed region starts.  N > 1 (launched by torch.distributed.run, one rank per GPU):
by default every rank proves its own independent traces (proofs are independent objects: no
data-path collective, weak scaling, `value` is the aggregate over ranks).  `--mode sharded` instead
splits ONE proof over the ranks (tap-stark_amd/csrc/sharded.cpp; needs N <= 2^log_blowup; RCCL
all-gathers of the trace, the Merkle sub-roots and the FRI tail) -- strong scaling, the latency mode
meant for BASELINE config 4 (`--workload config4`).

Prints ONE JSON line on rank 0 with the driver's contract fields plus:
  roofline     -- the dominant kernel's achieved algorithmic-bytes rate from HIP events recorded
                  around every launch on the library's own stream (a separate, untimed pass);
  cpu_baseline -- the CPU oracle (a port: the Rust reference cannot run here) on a bounded
                  sample of the same workload, on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def workload(name: str, log_n: int, need_host_trace: bool):
    """(air, host trace or None, public values, description, fri config, device-side generator).
    The traces are generated ON THE DEVICE (ts_trace_*): inputs are born in HBM; a host copy is only
    made when a mode needs to slice it."""
    import tapstark_amd as ts
    from tapstark_amd.airs import (FibonacciAir, SynthMulAir, generate_fibonacci_trace,
                                   generate_synth_mul_trace)
    n = 1 << log_n
    if name in ("config3", "config4"):
        air = SynthMulAir(64)
        trace = generate_synth_mul_trace(n) if need_host_trace else None
        pis = np.zeros(0, dtype=np.uint32)
        cfg = (2, 28, 8) if name == "config3" else (4, 16, 8)
        desc = (f"SynthMulAir-64 (build-defined), trace 2^{log_n}x64, log_blowup={cfg[0]}, "
                f"{cfg[1]} queries, pow 8")
        return air, trace, pis, desc, cfg, (n, 64), lambda c: ts.DeviceMatrix.synth_mul(c, n, 64)
    if name == "config5":  # build-defined stand-in for the RISC0-recursion-style AIR
        from tapstark_amd.airs import SynthExtAir, generate_synth_ext_trace
        air = SynthExtAir(163)
        host = generate_synth_ext_trace(n, 163) if need_host_trace else None
        pis = np.zeros(0, dtype=np.uint32)
        desc = (f"SynthExt-163 (build-defined, EF4 multiplication constraints), trace 2^{log_n}x163, "
                "log_blowup=4, 16 queries, pow 8")
        return air, host, pis, desc, (4, 16, 8), (n, 163), lambda c: ts.DeviceMatrix.synth_ext(c, n, 163)
    if name == "config2":
        air = FibonacciAir()
        trace = generate_fibonacci_trace(0, 1, n) if need_host_trace else None

        def last_right(c):  # pis = [a, b, trace[n-1].right]  (fib_air.rs:133-139)
            return int(ts.DeviceMatrix.fibonacci(c, 0, 1, n).download()[-1, 1])

        desc = f"Fibonacci AIR, trace 2^{log_n}x2, log_blowup=2, 28 queries, pow 8"
        return (air, trace, last_right, desc, (2, 28, 8), (n, 2),
                lambda c: ts.DeviceMatrix.fibonacci(c, 0, 1, n))
    raise SystemExit(f"unknown workload {name}")


def algorithmic_bytes_per_proof(n: int, w: int, b: int, qd: int) -> dict:
    """Algorithmic HBM bytes per proof, per kernel (DESIGN.md "Kernels and their rooflines"):
    each array counted once per launch that must read or write it."""
    N = n << b
    wall = w + 4 * qd  # every committed column (trace + quotient chunks)
    # Merkle parents (64 B read + 32 B written each): two N-leaf trees + FRI trees (N/2 + N/4 + ..).
    # Levels with > 2^17 children take one k_merkle_level launch each; the last 17 levels of a tree
    # are one k_merkle_tree launch; a FRI round with <= 2^17 leaves is one k_fri_round launch (fold
    # of the previous vector + leaf hashes + the whole tree), the rounds above keep fold / levels.
    TREE = 17
    parents_all = 3 * N  # N - 1 per N-leaf tree, and N/2 + N/4 + ... over the FRI trees
    fri_leaves = [N >> r for r in range(1, 40) if (N >> r) > 1024 // 2]
    lvl2 = lvl1 = 0
    for leaves in [N, N] + [h for h in fri_leaves if h > (1 << TREE)]:
        p = leaves // 2
        while p >= (1 << TREE):
            lvl1 += p
            p //= 2
    fused = [h for h in fri_leaves if h <= (1 << TREE)]
    fused_bytes = sum((64 + 8 + 32 + 32 + 96) * h for h in fused)  # prev, twiddles, cur, leaf digests, parents
    small = max(parents_all - lvl1 - lvl2 - sum(fused), 0)
    big_fri_elems = sum(2 * h for h in fri_leaves if h > (1 << TREE))
    return {
        "k_transpose_bitrev": 8 * n * w,
        "k_intt_contig": 8 * n * wall,
        "k_lde_mid<1>": 4 * n * wall + 4 * N * wall,
        "k_lde_mid<0>": 4 * n * wall + 4 * N * wall,
        "k_lde_fwd_contig": 8 * N * wall,
        # the trace (one matrix, width a multiple of 16) takes the strided kernel, the batch of
        # quotient chunks the pointer-table one; any other width sends both through the latter
        "k_leaf_hash_strided": 4 * N * w + 32 * N,
        "k_leaf_hash<2>": 4 * N * wall + 2 * 32 * N,
        "k_leaf_hash<1>": (4 * N * 4 * qd + 32 * N) if w % 16 == 0 else (4 * N * wall + 2 * 32 * N),
        "k_leaf_hash_ef_pairs": 16 * N + 16 * N,
        "k_merkle_level<2>": 96 * lvl2,
        "k_merkle_level<1>": 96 * lvl1,
        "k_merkle_tree": 96 * small,
        "k_fri_round<true>": fused_bytes,
        "k_selectors": 12 * n * qd,
        "k_quotient_jit": 4 * n * qd * w + 12 * n * qd + 16 * n * qd,
        "k_quotient<256>": 4 * n * qd * w + 12 * n * qd + 16 * n * qd,
        "k_bary_weights": 32 * n,
        "(k_bary_dots<2, 64>)": 4 * n * w + 32 * n,
        "(k_bary_dots<1, 8>)": qd * (16 * n + 16 * n),
        "k_reduce_fused": 4 * N * wall + 16 * N,
        "k_fri_fold_pairs": 16 * big_fri_elems + 8 * big_fri_elems + 8 * big_fri_elems,
    }


def _omp_set_threads(n: int):
    import ctypes
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def cpu_baseline(target_seconds: float = 20.0) -> dict:
    """The oracle prover (oracle/, a C port of the reference's algorithm: the Rust reference cannot
    be built here) on this box's host cores, on a bounded sample of the same workload: the median
    of 5 runs after a warm-up with every core of the GPU's share, plus one
    1-thread sample -- the reference as configured is single-threaded (no manifest enables
    p3-maybe-rayon's `parallel`)."""
    import tapstark_amd as ts
    from oracle import oracle_py as orc
    from tapstark_amd.airs import SynthMulAir, generate_synth_mul_trace

    # the GPU box gives one GPU's share of the host (16 cores); never oversubscribe
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("OMP_NUM_THREADS", min(avail, 16)))
    os.environ["OMP_NUM_THREADS"] = str(cores)  # read by libgomp when the oracle is first loaded
    air = SynthMulAir(64)
    tape = ts.air_tape(air, 0)
    cfg = orc.FriConfig(2, 28, 8)
    orc.prove(cfg, tape, generate_synth_mul_trace(1 << 8), [])  # thread-pool warm-up
    _omp_set_threads(cores)

    def timed(trace):
        t0 = time.perf_counter()
        orc.prove(cfg, tape, trace, [], cap_words=1 << 22)
        return time.perf_counter() - t0

    # size the sample from a probe: the largest 2^k <= 2^20 rows for which 1 warm-up + 5 runs fit
    probe = 15
    probe_trace = generate_synth_mul_trace(1 << probe)
    est = min(timed(probe_trace), timed(probe_trace))
    log_n = probe
    while log_n < 20 and est * 2 * 6 <= target_seconds:
        est *= 2
        log_n += 1
    trace = generate_synth_mul_trace(1 << log_n)
    timed(trace)  # warm-up at size
    runs = sorted(timed(trace) for _ in range(5))
    dt = runs[2]
    # one thread, on a sample 1/8 the size (about the same wall time)
    log_n1 = max(log_n - 3, 10)
    trace1 = generate_synth_mul_trace(1 << log_n1)
    _omp_set_threads(1)
    dt1 = min(timed(trace1), timed(trace1))
    _omp_set_threads(cores)
    return {"value": (64 << log_n) / dt, "unit": "trace cells/sec", "cores": cores, "kind": "port",
            "sample": f"oracle prove() of SynthMulAir-64 2^{log_n}x64, log_blowup=2, 28 queries: median of 5 "
                      f"runs after a warm-up ({runs[0]:.2f}..{runs[4]:.2f} s, median {dt:.2f} s), OpenMP over "
                      f"{cores} host threads",
            "proofs_per_sec": 1.0 / dt,
            "runs_s": [round(r, 3) for r in runs],
            "one_thread": {"value": (64 << log_n1) / dt1, "unit": "trace cells/sec", "cores": 1,
                           "sample": f"the same prover on 2^{log_n1}x64, best of 2 ({dt1:.2f} s), 1 thread"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 48 steps = 12 per lane: the lanes start in lockstep and only drift into complementary phases
    # after a few proofs (K = 12: 3.3-3.4 ms/step, K = 40: 3.0)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="config3", choices=["config3", "config2", "config4", "config5"])
    ap.add_argument("--log-n", type=int, default=None, help="default 20 (22 for config4)")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded"],
                    help="N > 1: independent proofs per GPU (default) or one proof over all GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-traces", action="store_true",
                    help="hand every step its trace as a HOST buffer (ts_matrix_upload inside the "
                         "timed region): the PCIe-inclusive rate, reported in DESIGN.md, never `value`")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TS_BENCH_STREAMS", "4")),
                    help="independent proofs in flight per GPU (one context = one HIP stream and one "
                         "host thread each); the K timed steps are shared among them")
    args = ap.parse_args()
    # Native libraries print to stdout on their own (RCCL's version banner at communicator creation):
    # everything but the one JSON line goes to stderr, the line itself to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if args.log_n is None:
        args.log_n = 22 if args.workload == "config4" else 20
    sharded = args.mode == "sharded"

    from tapstark_amd.benchutil import init_dist, run_timed

    env = init_dist()
    if env.world != args.gpus and env.world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env.world}")

    import tapstark_amd as ts
    from tapstark_amd.build import build

    if not os.path.exists(ts._lib.LIB_PATH):  # normally built by __graft_entry__.build() beforehand
        if env.rank == 0:
            build()
        if env.dist is not None:
            env.dist.barrier()
    # one GPU per rank (TS_BENCH_SHARE_GPU=1 lets a rehearsal put every rank on GPU 0)
    dev = 0 if os.environ.get("TS_BENCH_SHARE_GPU") else env.local_rank
    ctx = ts.Context(dev)  # raises without a GPU: there is no fallback path

    air, trace, pis, desc, cfg, (n, w), make_trace = workload(args.workload, args.log_n, False)
    if callable(pis):
        pis = np.array([0, 1, pis(ctx)], dtype=np.uint32)
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), ctx))
    cair = ts.CompiledAir(ctx, ts.air_tape(air, len(pis)))
    qd = 1 << cair.log_quotient_degree

    # Proofs are independent, so S of them are kept in flight per GPU: S contexts (one HIP stream,
    # one device pool and one host thread each).  The serial transcript of one proof leaves the
    # GPU idle at its host round trips; the next proof's kernels fill those gaps.
    # The K timed steps are dealt to the lanes in advance, so K should be a multiple of S: if the
    # requested lane count does not divide K, a neighbouring one that does is used (K = 10 -> 5 lanes).
    S = 1 if sharded else max(1, min(args.streams, args.steps))
    if S > 1 and args.steps % S:
        S = next((c for c in (S + 1, S - 1, S + 2, S - 2) if c >= 2 and args.steps % c == 0), S)
    lanes = [(ctx, config, cair)]
    for _ in range(1, S):
        c2 = ts.Context(dev)
        lanes.append((c2, ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*cfg), c2)),
                      ts.CompiledAir(c2, ts.air_tape(air, len(pis)))))

    # inputs resident in HBM before the timed region (prove() consumes its trace, like the
    # reference's moved RowMajorMatrix, so one copy per step); step i runs on lane i % S
    total = args.warmup + args.steps
    last = {}
    if sharded:
        # one proof per step over all ranks: rank g is handed natural rows [g n/G, (g+1) n/G)
        import torch
        import torch.distributed as dist
        from tapstark_amd.dist import TorchComm

        if env.dist is None:  # a world of one still goes through the RCCL callbacks
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29512", rank=0, world_size=1,
                                    device_id=torch.device("cuda", dev))
        # a proof can be split over at most 2^log_blowup ranks (whole cosets per rank): with more
        # GPUs than that, groups of that size each prove their own proofs (TS_BENCH_GROUP overrides
        # the group size for rehearsals)
        gsize = min(env.world, 1 << cfg[0], int(os.environ.get("TS_BENCH_GROUP", env.world)))
        while env.world % gsize:
            gsize //= 2
        n_groups = env.world // gsize
        group = None
        if n_groups > 1:
            for gi in range(n_groups):  # every rank takes part in creating every group
                gr = dist.new_group(list(range(gi * gsize, (gi + 1) * gsize)))
                if env.rank // gsize == gi:
                    group = gr
        grank = env.rank % gsize
        # the collectives: the library's own RCCL communicator (csrc/comm.cpp: ncclAllGather /
        # ncclBroadcast on the context's stream; what a Rust host would link), bootstrapped with a
        # unique id made by each group's first rank and handed round over torch.distributed;
        # TS_BENCH_COMM=torch uses the torch.distributed callbacks (tap-stark_amd/dist.py) instead
        use_native = os.environ.get("TS_BENCH_COMM", "rccl") == "rccl" and dist.get_backend() == "nccl"
        if use_native:
            from tapstark_amd import comm as tcomm
            ids = [None] * dist.get_world_size()
            dist.all_gather_object(ids, tcomm.rccl_unique_id() if grank == 0 else None)
            comm = tcomm.RcclComm(ctx, ids[(env.rank // gsize) * gsize], grank, gsize)
            comm.backend = "rccl (native ts_comm)"
        else:
            comm = TorchComm(dev, group=group)
        # every rank generates the whole trace on its own device (ts_trace_*): nothing to exchange
        # for the input; TS_BENCH_SLICED=1 hands out row slices instead (adds the trace all-gather)
        sliced = bool(os.environ.get("TS_BENCH_SLICED"))
        if sliced:
            full = make_trace(ctx).download()
            rows = np.ascontiguousarray(full[grank * n // gsize:(grank + 1) * n // gsize])
            mats = [ts.DeviceMatrix.upload(ctx, rows) for _ in range(total + 1)]
        else:
            mats = [make_trace(ctx) for _ in range(total + 1)]

        def prove_one(i):
            last["proof"] = ts.prove_sharded(config, cair, ts.BfChallenger(), mats[i], pis, comm,
                                             trace_replicated=not sliced)
    else:
        # one resident trace per step (prove() consumes it); beyond 100 GB of them (288 GB of HBM) the
        # trace of a step is generated on the device at the start of the step instead, inside the
        # timed region (measured: 3.4 instead of 3.0 ms/step)
        pregen = total * n * w * 4 <= 100 << 30
        mats = [make_trace(lanes[i % S][0]) for i in range(total)] if pregen else None

        if args.host_traces:
            pregen, mats = False, None
            host_trace = make_trace(ctx).download()

        def prove_one(i):
            c, conf, ca = lanes[i % S]
            if args.host_traces:
                m = ts.DeviceMatrix.upload(c, host_trace)
            else:
                m = mats[i] if pregen else make_trace(c)
            last["proof"] = ts.prove(conf, ca, ts.BfChallenger(), m, pis)
            if pregen:
                mats[i] = None  # the matrix handle is spent

    if S == 1:
        step = prove_one
        run_steps = None
    else:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=S)
        step = prove_one

        def run_steps(first, count):
            # lane l proves steps first+l, first+l+S, ... in its own thread (ctypes drops the GIL)
            def lane_job(l):
                for i in range(first + l, first + count, S):
                    prove_one(i)
            list(pool.map(lane_job, range(S)))

    def local_sync():
        for c, _, _ in lanes:
            c.synchronize()
        if env.dist is not None and env.device is not None:
            import torch
            torch.cuda.synchronize()

    # Prime every lane with one untimed proof of its own (beyond the W warm-up steps, which only
    # reach the first W lanes): a context builds its twiddle / scale / selector tables and grows its
    # device pool on first use, and that must not fall into the timed region.
    if not sharded:
        prime = [make_trace(lanes[l][0]) for l in range(S)]

        def prime_lane(l):
            c, conf, ca = lanes[l]
            ts.prove(conf, ca, ts.BfChallenger(), prime[l], pis)
        if S == 1:
            prime_lane(0)
        else:
            list(pool.map(prime_lane, range(S)))
        del prime

    # sharded: the ranks of a group share each step's n*w cells
    res = run_timed(env, step, args.steps, args.warmup, local_sync,
                    units_per_step=float(n * w) / (gsize if sharded else 1), run_steps=run_steps)
    shard_stages = None
    if sharded:
        res["steps_per_sec"] = n_groups * args.steps / res["elapsed_s"]
        # one more proof with the stage timers on, on every rank: where a rank's time goes
        ctx.set_timing(True)
        prove_one(total)
        mine = {}
        for k, v in ctx.take_timings():
            mine[k] = round(mine.get(k, 0.0) + v, 3)
        ctx.set_timing(False)
        shard_stages = [None] * dist.get_world_size()
        dist.all_gather_object(shard_stages, mine)

    out = None
    if env.rank == 0:
        # ---- roofline leg: per-kernel HIP-event timings over 3 extra (untimed) proofs
        reps = 3
        extra = [make_trace(ctx) for _ in range(reps)]
        ctx.set_kernel_timing(True)
        for m in extra:
            ts.prove(config, cair, ts.BfChallenger(), m, pis)
        kt = ctx.take_kernel_timings()
        ctx.set_kernel_timing(False)
        ctx.set_timing(True)
        ts.prove(config, cair, ts.BfChallenger(), make_trace(ctx), pis)
        stages = ctx.take_timings()
        ctx.set_timing(False)
        # one proof alone on the GPU, no timers inside (the stage timers synchronise at every stage
        # boundary): host wall clock around ts_prove, which returns with the proof on the host
        lat = []
        for _ in range(5):
            m_ = make_trace(ctx)
            ctx.synchronize()
            t0 = time.perf_counter()
            ts.prove(config, cair, ts.BfChallenger(), m_, pis)
            lat.append(1e3 * (time.perf_counter() - t0))
        single_latency = sorted(lat)[len(lat) // 2]
        stage_sum = {}
        for k, v in stages:  # a stage name can occur twice (trace commit, quotient commit)
            stage_sum[k] = round(stage_sum.get(k, 0.0) + v, 3)
        alg = algorithmic_bytes_per_proof(n, w, cfg[0], qd)

        def alg_bytes(name):  # every instantiation of the strided NTT pass moves the same bytes
            if "k_lde_mid" in name:
                return alg["k_lde_mid<1>"]
            if name == "k_merkle_level<1>" and "k_merkle_level<2>" not in kt:
                return alg["k_merkle_level<1>"] + alg["k_merkle_level<2>"]
            return alg.get(name)

        per_kernel = {}
        for name, (cnt, ms) in kt.items():
            ms_pp = ms / reps
            b = alg_bytes(name)
            per_kernel[name] = {
                "launches_per_proof": cnt / reps, "ms_per_proof": round(ms_pp, 4),
                "avg_launch_ms": round(ms / cnt, 5),
                "alg_gbps": round(b / (ms_pp * 1e-3) / 1e9, 1) if b and ms_pp > 0 else None}
        dom = max(kt.items(), key=lambda kv: kv[1][1])[0]
        dom_ms_pp = kt[dom][1] / reps
        achieved = (alg_bytes(dom) or 0) / (dom_ms_pp * 1e-3)
        # HBM traffic of that kernel from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled for gfx950:
        # profiles/*_pmc_traffic.json, made from tools/prof_prove.py); bytes per launch, or null
        # NOT measured in this run (PMC collection needs rocprofv3 around the process): the file it
        # comes from and that file's hash are reported beside it
        traffic, traffic_source = None, None
        try:
            import glob
            import hashlib
            tag = {"config3": "", "config2": "config2_", "config4": "config4_"}.get(args.workload)
            pmc_files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))
                               if tag is not None and (("config" in os.path.basename(f)) == bool(tag))
                               and (not tag or tag in os.path.basename(f)))
            if pmc_files and args.log_n == (22 if args.workload == "config4" else 20):
                ks = json.load(open(pmc_files[-1]))["kernels"]
                stem = dom.strip("()").rstrip(">")  # "k_lde_mid<1" matches "k_lde_mid<1, 8192, 512>"
                pk = ks.get(dom) or next((v for k, v in ks.items() if k.startswith(stem)), None)
                if pk:
                    traffic = round((pk["fetch_bytes_per_proof_corrected"] + pk["write_bytes_per_proof"])
                                    / pk["launches_per_proof"])
                    traffic_source = {
                        "file": os.path.relpath(pmc_files[-1], ROOT),
                        "sha256": hashlib.sha256(open(pmc_files[-1], "rb").read()).hexdigest(),
                        "note": "static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run "
                                "(tools/prof_prove.py), not collected in this run"}
        except Exception:
            traffic, traffic_source = None, None
        lpp = kt[dom][0] / reps
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved / 1e9, 2),
                    "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "alg_bytes_per_launch": round((alg_bytes(dom) or 0) / lpp) if lpp else None,
                    "avg_launch_ms": round(kt[dom][1] / kt[dom][0], 5),
                    "launches_per_proof": kt[dom][0] / reps,
                    "alg_bytes_per_proof": alg_bytes(dom),
                    "kernel_ms_total_per_proof": round(sum(v[1] for v in kt.values()) / reps, 3)}
        # ---- the stage and the whole job against the same roofline (SURVEY.md section 8(d) terms)
        N_ = n << cfg[0]
        wall_ = w + 4 * qd
        A1 = 4 * n * w + 4 * N_ * w            # trace LDE: read n x w, write N x w
        C1 = 16 * n * qd + 16 * N_ * qd        # chunk LDE
        whole = (A1 + 32 * (2 * N_ - 1) + (4 * n * qd * w + 16 * n * qd) + C1 + 32 * (2 * N_ - 1)
                 + (4 * N_ * w + 16 * N_ * qd + 16 * N_) + (32 * N_ + 16 * N_ + 64 * N_))
        lde_ms = stage_sum.get("coset_lde")
        roofline_stage = None
        if lde_ms:
            roofline_stage = {"stage": "coset_lde (trace + quotient chunks)", "alg_bytes": A1 + C1,
                              "ms": lde_ms, "achieved": round((A1 + C1) / (lde_ms * 1e-3) / 1e9, 1),
                              "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                              "frac": round((A1 + C1) / (lde_ms * 1e-3) / HBM_PEAK, 4),
                              "note": "single proof alone on the GPU; the stage is three passes (60 n W bytes "
                                      "moved for 20 n W algorithmic) and VALU-bound, see alu_ceiling"}
        roofline_whole = {"alg_bytes_per_proof": whole, "ms_per_step": round(res["ms_per_step"], 4),
                          "achieved": round(whole / (res["ms_per_step"] * 1e-3) / 1e9, 1), "peak": HBM_PEAK / 1e9,
                          "unit": "GB/s", "frac": round(whole / (res["ms_per_step"] * 1e-3) / HBM_PEAK, 4)}
        # ---- the integer-ALU ceiling, measured in this run with the library's own arithmetic
        log_n_ = n.bit_length() - 1
        butterflies = wall_ * (n // 2) * log_n_ * (1 + (1 << cfg[0]))
        compressions = N_ * ((4 * w + 63) // 64) + N_ * ((16 * qd + 63) // 64) + N_ + 3 * N_

        def ms_of(*names):
            return sum(v["ms_per_proof"] for k, v in per_kernel.items() if any(x in k for x in names))
        bf_peak, b3_peak = ctx.alu_ceiling(0), ctx.alu_ceiling(1)
        ntt_ms = ms_of("k_intt_contig", "k_lde_mid", "k_lde_fwd_contig")
        b3_ms = ms_of("k_leaf_hash", "k_merkle_level", "k_merkle_tree", "k_fri_round")
        alu_ceiling = {
            "butterflies_per_s_peak": round(bf_peak), "butterflies_per_proof": butterflies,
            "ntt_kernels_ms": round(ntt_ms, 4), "ntt_ms_at_peak": round(butterflies / bf_peak * 1e3, 4),
            "ntt_frac_of_alu_peak": round(butterflies / bf_peak * 1e3 / ntt_ms, 4) if ntt_ms else None,
            "blake3_compressions_per_s_peak": round(b3_peak), "blake3_compressions_per_proof": compressions,
            "merkle_kernels_ms": round(b3_ms, 4), "merkle_ms_at_peak": round(compressions / b3_peak * 1e3, 4),
            "merkle_frac_of_alu_peak": round(compressions / b3_peak * 1e3 / b3_ms, 4) if b3_ms else None,
            "note": "peaks from ts_bench_alu (register-resident loops of the same butterfly / compression code, "
                    "no memory traffic); FRI-round leaf hashes fused into the fold kernel are not in merkle_kernels_ms"}
        # ---- the ceiling that actually binds: VALU instruction issue.  A wave64 VALU instruction
        # holds its SIMD for four cycles (tools/pmc_alu.sh: the register-resident loops issue exactly
        # one per 4 cycles per SIMD at 2.30-2.34 GHz); the prover's kernels run at ~2.0 GHz
        # (SQ_BUSY_CYCLES over their durations).  SQ_INSTS_VALU summed over one proof comes from a
        # committed rocprofv3 --pmc pass (static, like roofline.traffic): tools/pmc_sq.sh.
        valu_issue = None
        try:
            import hashlib
            import re as _re
            sqf = os.path.join(ROOT, "profiles", f"r02_{args.workload}_sq_counters.txt")
            if os.path.exists(sqf) and args.log_n == (22 if args.workload == "config4" else 20):
                m_ = _re.search(r"whole proof: SQ_INSTS_VALU ([0-9.e+]+)", open(sqf).read())
                insts = float(m_.group(1))
                simds, clock = 4 * ctx.num_cus if hasattr(ctx, "num_cus") else 1024, 2.0e9
                floor_ms = insts * 4 / simds / clock * 1e3
                valu_issue = {
                    "wave_instructions_per_proof": insts, "simds": simds, "cycles_per_instruction": 4,
                    "clock_hz_under_load": clock, "ms_per_proof_at_ceiling": round(floor_ms, 4),
                    "frac_of_ceiling": round(floor_ms / res["ms_per_step"], 4),
                    "source": {"file": os.path.relpath(sqf, ROOT),
                               "sha256": hashlib.sha256(open(sqf, "rb").read()).hexdigest(),
                               "note": "static: SQ_INSTS_VALU of an earlier rocprofv3 --pmc pass, not collected "
                                       "in this run; clock = SQ_BUSY_CYCLES / kernel durations of the same pass"}}
        except Exception:
            valu_issue = None
        # ---- the rate a caller sees who hands over HOST traces (never `value`): pinned buffer,
        # asynchronous upload on each lane's stream, the upload of one lane overlapping the proofs of the others
        h2d = None
        if env.world == 1 and not sharded and not args.host_traces and args.workload in ("config3", "config2") and S > 1:
            try:
                pin = ts.PinnedHostMatrix(n, w)
                pin.array[:] = make_trace(ctx).download()
                k2 = 6 * S

                def h2d_job(l):
                    c, conf, ca = lanes[l]
                    for _ in range(k2 // S):
                        ts.prove(conf, ca, ts.BfChallenger(), ts.DeviceMatrix.upload_async(c, pin), pis)
                list(pool.map(h2d_job, range(S)))  # warm-up
                local_sync()
                t0 = time.perf_counter()
                list(pool.map(h2d_job, range(S)))
                local_sync()
                dt_h = time.perf_counter() - t0
                h2d = {"ms_per_step": round(1e3 * dt_h / k2, 4), "steps": k2,
                       "h2d_GB_per_s": round(n * w * 4 * k2 / dt_h / 1e9, 1),
                       "note": "every step uploads its trace from page-locked host memory inside the timed "
                               "region (hipMemcpyAsync on the lane's stream); PCIe Gen5 x16 bounds it"}
            except Exception as e:  # never let the extra leg take the headline down
                h2d = {"error": repr(e)}
        # (the contract: timed on rank 0 at N = 1 only)
        cpu = None if (args.no_cpu_baseline or env.world > 1) else cpu_baseline()
        proof = last["proof"]
        out = {
            "metric": "trace cells/sec (proofs/sec alongside), 2^20x64 BabyBear trace",
            "value": res["value"], "unit": "trace cells/sec", "n_gpus": env.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": ("strong" if n_groups == 1 else "strong within a group, weak across groups")
                       if sharded else "weak", "vs_baseline": None,
            "dtype": "u32",  # BabyBear arithmetic on u32 lanes (64-bit intermediates)
            "data": ("synthetic (trace uploaded from host memory INSIDE the timed region: PCIe-inclusive)"
                     if args.host_traces else
                     "synthetic (trace generated on the device, resident in HBM before the timed region)"),
            "config": {"workload": desc, "rows": n, "width": w, "log_blowup": cfg[0],
                       "num_queries": cfg[1], "proof_of_work_bits": cfg[2], "quotient_degree": qd,
                       "parallelism": (f"{n_groups} group(s) of {gsize} GPU(s), one proof sharded over each group, "
                                        f"collectives over {comm.backend}" if sharded else
                                       ("1 rank per GPU (replicas)" if env.world > 1 else "1 GPU")
                                       + f", {S} proofs in flight per GPU"),
                       "quotient_kernel": "hiprtc-specialised" if cair.is_jit else "interpreter",
                       "proof_words": int(len(proof.words))},
            "proofs_per_sec": res["steps_per_sec"],
            # one proof alone on the GPU, HIP events around ts_prove (the `value` above keeps
            # several in flight; this is the latency a single caller sees)
            "single_proof_latency_ms": round(single_latency, 4),
            "single_proof_latency_with_stage_timers_ms": stage_sum.get("prove"),
            "roofline": roofline, "roofline_stage": roofline_stage, "roofline_whole": roofline_whole,
            "alu_ceiling": alu_ceiling, "valu_issue": valu_issue, "h2d_inclusive": h2d,
            "h2d_inclusive_ms_per_step": (h2d or {}).get("ms_per_step"), "cpu_baseline": cpu,
            "stages_ms": stage_sum,
            "shard_stages_ms_per_rank": shard_stages,
            "kernels": per_kernel,
        }
    if sharded and env.dist is None:
        import torch.distributed as dist
        dist.destroy_process_group()
    env.close()
    if out is not None:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
