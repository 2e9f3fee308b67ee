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
        host = generate_synth_ext_trace(n, 163)  # no device generator for this one: uploaded
        pis = np.zeros(0, dtype=np.uint32)
        desc = (f"SynthExt-163 (build-defined, EF4 multiplication constraints), trace 2^{log_n}x163, "
                "log_blowup=4, 16 queries, pow 8")
        return air, host, pis, desc, (4, 16, 8), (n, 163), lambda c: ts.DeviceMatrix.upload(c, host)
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
    fri_elems = 2 * N  # sum over rounds of the folded vector lengths (N + N/2 + ...)
    # Merkle parents (64 B read + 32 B written each): two N-leaf trees + FRI trees (N/2 + N/4 + ..);
    # levels with > 2^16 children go through k_merkle_level, the rest through the subtree kernel
    parents_all = 3 * N  # N - 1 per N-leaf tree, and N/2 + N/4 + ... over the FRI trees
    # per-level launches take the levels with >= 2^16 parents (two parents per thread from 2^18 up)
    lvl2 = lvl1 = 0
    for leaves in [N, N] + [N >> r for r in range(1, 40) if (N >> r) > (1 << 16)]:
        p = leaves // 2
        while p >= (1 << 16):
            if p >= (1 << 18):
                lvl2 += p
            else:
                lvl1 += p
            p //= 2
    small = max(parents_all - lvl1 - lvl2, 0)
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
        "(k_merkle_subtree<NTH, LOG_S>)": 96 * small,
        "k_selectors": 12 * n * qd,
        "k_quotient_jit": 4 * n * qd * w + 12 * n * qd + 16 * n * qd,
        "k_quotient<256>": 4 * n * qd * w + 12 * n * qd + 16 * n * qd,
        "k_bary_weights": 32 * n,
        "(k_bary_dots<2, 64>)": 4 * n * w + 32 * n,
        "(k_bary_dots<1, 8>)": qd * (16 * n + 16 * n),
        "k_reduce_fused": 4 * N * wall + 16 * N,
        "k_fri_fold_pairs": 16 * fri_elems + 8 * fri_elems + 8 * fri_elems,
    }


def cpu_baseline(target_seconds: float = 30.0) -> dict:
    """The oracle prover (oracle/, a C port of the reference's algorithm) on this box's host
    cores, on a bounded sample of the same workload."""
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
    # size the sample from a probe: the largest 2^k <= 2^20 rows expected to take <= target_seconds
    probe = 16
    probe_trace = generate_synth_mul_trace(1 << probe)
    est = 1e9
    for _ in range(2):  # the first call also pays first-touch costs; keep the faster one
        t0 = time.perf_counter()
        orc.prove(cfg, tape, probe_trace, [], cap_words=1 << 22)
        est = min(est, time.perf_counter() - t0)
    log_n = probe
    while log_n < 20 and est * 2 <= target_seconds:
        est *= 2
        log_n += 1
    trace = generate_synth_mul_trace(1 << log_n)
    t0 = time.perf_counter()
    orc.prove(cfg, tape, trace, [], cap_words=1 << 22)
    dt = time.perf_counter() - t0
    return {"value": (64 << log_n) / dt, "unit": "trace cells/sec", "cores": cores, "kind": "port",
            "sample": f"one oracle prove() of SynthMulAir-64 2^{log_n}x64, log_blowup=2, 28 queries "
                      f"({dt:.2f} s, OpenMP over {cores} host threads)",
            "proofs_per_sec": 1.0 / dt}


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
        comm = TorchComm(dev, group=group)
        grank = env.rank % gsize
        # every rank generates the whole trace on its own device (ts_trace_*): nothing to exchange
        # for the input; TS_BENCH_SLICED=1 hands out row slices instead (adds the trace all-gather)
        sliced = bool(os.environ.get("TS_BENCH_SLICED"))
        if sliced:
            full = make_trace(ctx).download()
            rows = np.ascontiguousarray(full[grank * n // gsize:(grank + 1) * n // gsize])
            mats = [ts.DeviceMatrix.upload(ctx, rows) for _ in range(total)]
        else:
            mats = [make_trace(ctx) for _ in range(total)]

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
    if sharded:
        res["steps_per_sec"] = n_groups * args.steps / res["elapsed_s"]

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
        traffic = None
        try:
            import glob
            pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
            if pmc_files and args.workload == "config3" and args.log_n == 20:  # what was profiled
                ks = json.load(open(pmc_files[-1]))["kernels"]
                stem = dom.strip("()").rstrip(">")  # "k_lde_mid<1" matches "k_lde_mid<1, 8192, 512>"
                pk = ks.get(dom) or next((v for k, v in ks.items() if k.startswith(stem)), None)
                if pk:
                    traffic = round((pk["fetch_bytes_per_proof_corrected"] + pk["write_bytes_per_proof"])
                                    / pk["launches_per_proof"])
        except Exception:
            traffic = None
        lpp = kt[dom][0] / reps
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved / 1e9, 2),
                    "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                    "traffic": traffic,
                    "alg_bytes_per_launch": round((alg_bytes(dom) or 0) / lpp) if lpp else None,
                    "avg_launch_ms": round(kt[dom][1] / kt[dom][0], 5),
                    "launches_per_proof": kt[dom][0] / reps,
                    "alg_bytes_per_proof": alg_bytes(dom),
                    "kernel_ms_total_per_proof": round(sum(v[1] for v in kt.values()) / reps, 3)}
        cpu = None if args.no_cpu_baseline else cpu_baseline()
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
            "single_proof_latency_ms": stage_sum.get("prove"),
            "roofline": roofline, "cpu_baseline": cpu,
            "stages_ms": stage_sum,
            "kernels": per_kernel,
        }
    if sharded and env.dist is None:
        import torch.distributed as dist
        dist.destroy_process_group()
    env.close()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
