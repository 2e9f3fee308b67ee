"""The N > 1 path of bench.py (one rank per GPU, barrier, max-over-ranks time, aggregate value)
rehearsed on CPU with the gloo backend at world_size 2.  The prover itself needs a GPU, so the
step here is a stand-in; what is covered is the distributed timing protocol that bench.py uses
unchanged (tapstark_amd.benchutil)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from tapstark_amd.benchutil import init_dist, run_timed
    env = init_dist(backend="gloo")
    assert env.world == 2
    calls = []
    def step(i):
        calls.append(i)
        time.sleep(0.02 * (1 + env.rank))      # rank 1 is twice as slow: the job time is ITS time
    res = run_timed(env, step, steps=3, warmup=1, local_sync=lambda: None, units_per_step=100.0)
    assert calls == [0, 1, 2, 3]
    print(json.dumps({"rank": env.rank, **res}), flush=True)
    env.close()
""") % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_timing_protocol(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        outs.append(o)
    import json

    res = [json.loads(o.strip().splitlines()[-1]) for o in outs]
    # both ranks agree on the job-level numbers
    assert abs(res[0]["elapsed_s"] - res[1]["elapsed_s"]) < 1e-9
    assert abs(res[0]["value"] - res[1]["value"]) < 1e-6
    # the slow rank (0.04 s per step) sets the time; value aggregates BOTH ranks' units
    assert res[0]["elapsed_s"] >= 3 * 0.04 * 0.9
    total_units = 2 * 3 * 100.0
    assert abs(res[0]["value"] - total_units / res[0]["elapsed_s"]) < 1e-6
    assert abs(res[0]["steps_per_sec"] - 6 / res[0]["elapsed_s"]) < 1e-6


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from tapstark_amd.benchutil import DistEnv, run_timed

    env = DistEnv(0, 0, 1)
    n = []
    res = run_timed(env, lambda i: n.append(i), steps=4, warmup=2, local_sync=lambda: None,
                    units_per_step=10.0)
    assert n == [0, 1, 2, 3, 4, 5] and res["value"] > 0


def _run_bench(args, extra_env, timeout=300):
    env = dict(os.environ, TS_BENCH_STUB="1", TS_BENCH_BACKEND="gloo", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        if k not in extra_env:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_as_a_plain_command_launches_two_ranks():
    # `bench.py --gpus N` with WORLD_SIZE unset must not silently run on one GPU.
    # The parent spawns the ranks itself (before any GPU call), relays rank 0's line, and that line
    # says n_gpus = 2.  TS_BENCH_STUB replaces the prover by a sleep so that this runs without a GPU.
    import json

    r = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 2 and rec["stub"] is True and rec["self_launched"] is True
    assert rec["steps"] == 4 and len(rec["windows_ms_per_step"]) == 3
    assert abs(rec["value"] - 2 * 4 / (rec["ms_per_step"] * 4e-3)) < 1e-6 * rec["value"]  # both ranks' steps


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    r = _run_bench(["--gpus", "8", "--steps", "4", "--warmup", "1"],
                   {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(_free_port())}, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    assert r.stdout.strip() == ""
    # and --gpus 1 inside a 2-rank launch is refused as well (the escape hatch VERDICT r2 named)
    r = _run_bench(["--gpus", "1", "--steps", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                      "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())},
                   timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_bench_self_launch_propagates_a_failing_rank():
    # a rank that dies takes the job down with a non-zero exit and no JSON line
    r = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1"], {"TS_BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_bench_without_gpu_fails_loudly():
    # no stub, no GPU in this container: the product path must raise, not fall back
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TS_BENCH_STUB"):
        env.pop(k, None)
    import torch

    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_bench_under_torch_distributed_run():
    # the driver's own launch line for N > 1 (one rank per GPU): the process IS a rank, no self-launch
    import json

    env = dict(os.environ, TS_BENCH_STUB="1", TS_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["self_launched"] is False


def test_stub_line_names_both_sharded_blocks():
    import json

    r = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["sharded_config4"] == {"skipped": "stub"} and rec["sharded_config5"] == {"skipped": "stub"}


def test_collectives_table_from_stage_timings():
    sys.path.insert(0, ROOT)
    from tapstark_amd.benchutil import split_stage_timings

    items = [("coset_lde", 1.5), ("collective: all_gather 32 B/rank (commit sub-roots)", 0.02),
             ("merkle_commit", 0.5), ("coset_lde", 0.25),
             ("collective: all_gather 32 B/rank (commit sub-roots)", 0.04),
             ("collective: broadcast 67108864 B (quotient chunk)", 1.0),
             ("collective: all_gather 32 B/rank (FRI round sub-roots)", 0.03)]
    stages, table = split_stage_timings(items)
    assert stages == {"coset_lde": 1.75, "merkle_commit": 0.5}
    rows = {r["what"]: r for r in table}
    assert rows["all_gather 32 B/rank (commit sub-roots)"]["count"] == 2
    assert abs(rows["all_gather 32 B/rank (commit sub-roots)"]["ms_total"] - 0.06) < 1e-9
    assert rows["all_gather 32 B/rank (commit sub-roots)"]["ms_max"] == 0.04
    assert rows["broadcast 67108864 B (quotient chunk)"]["bytes"] == 67108864
    assert rows["broadcast 67108864 B (quotient chunk)"]["collective"] == "broadcast"
    assert len(table) == 3


def test_shared_gpu_rehearsal_refuses_nccl():
    # with every rank on GPU 0 RCCL fails with "Duplicate GPU detected"; say so up front
    import pytest

    sys.path.insert(0, ROOT)
    from tapstark_amd import benchutil

    old = dict(os.environ)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", TS_BENCH_SHARE_GPU="1")
    try:
        with pytest.raises(SystemExit, match="TS_BENCH_BACKEND=gloo"):
            benchutil.init_dist(backend="nccl")
    finally:
        os.environ.clear()
        os.environ.update(old)


def test_watchdog_fires_on_its_thread_and_names_the_phase():
    import time

    sys.path.insert(0, ROOT)
    import bench

    seen = []
    wd = bench.Watchdog(0.05, seen.append)
    wd.phase = "sharded_config4: replicated: timed proofs"
    time.sleep(0.3)
    assert seen == ["sharded_config4: replicated: timed proofs"]
    quiet = []
    wd2 = bench.Watchdog(0.05, quiet.append)
    wd2.done()
    time.sleep(0.2)
    assert quiet == []
    # the timeout handler of the N > 1 block leaves with a NON-zero code (a hang must be visible)
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def on_timeout(phase):"):src.index("env.dist.barrier()  # rank 0 comes here later")]
    assert "os._exit(3)" in body and "os._exit(0)" not in body
    assert "except BaseException" not in src[src.index("# ---- N > 1: BASELINE configs 4 and 5"):]


# ------------------------------------------------------------------ dry-run of the 8-GPU lease on a CPU
def _run_bench_stub_lib(args, extra_env=None, timeout=600, launcher=None):
    """bench.py's REAL path (lanes, gate, priming, windows, record, rank-0 legs, sharded blocks) against
    tests/stub_lib (objects that sleep), gloo for the barrier / reductions."""
    env = dict(os.environ, TS_BENCH_STUB_LIB=os.path.join(ROOT, "tests", "stub_lib"), TS_BENCH_BACKEND="gloo",
               TS_BENCH_LIVE_PMC="0", TS_BENCH_PRIME_S="0.2", TS_STUB_STEP_S="0.003", TS_BENCH_SHARD_STEPS="2",
               TS_BENCH_SAMPLER="0", **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE", "TS_BENCH_STUB"):
        env.pop(k, None)
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py"), *args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def _check_contract_line(rec, n_gpus, steps):
    import math

    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "windows_ms_per_step", "windows", "priming"):
        assert key in rec, key
    assert rec["n_gpus"] == n_gpus and rec["steps"] == steps and rec["scaling"] == "weak"
    assert rec["unit"] == "trace cells/sec" and rec["dtype"] == "u32" and rec["vs_baseline"] is None
    assert rec["stub_lib"] is True and "STUB LIBRARY" in rec["data"]
    assert "model" not in rec["config"] and rec["config"]["rows"] == 1 << 20 and rec["config"]["width"] == 64
    # value = the cells ALL ranks proved / the slowest rank's time; ms_per_step is that time / K
    cells = n_gpus * steps * float((1 << 20) * 64)
    assert math.isclose(rec["value"], cells / (rec["ms_per_step"] * steps * 1e-3), rel_tol=1e-9)
    assert math.isclose(rec["proofs_per_sec"], n_gpus * steps / (rec["ms_per_step"] * steps * 1e-3), rel_tol=1e-9)
    assert math.isclose(rec["windows_ms_per_step"][0], rec["ms_per_step"], rel_tol=1e-3)
    assert len(rec["windows"]) == len(rec["windows_ms_per_step"]) == 3
    assert rec["priming"]["probes_ms_per_step"], "the priming probes are on the record"


def test_dry_run_of_the_eight_gpu_lease():
    """`python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 ...` (the driver's line) with the
    stub library: the record rank 0 prints is the one the driver will parse -- n_gpus 8, value = sum of
    cells / max time -- and both sharded blocks ran over ONE group of 8 through the native-communicator
    branch (unique id made by rank 0, handed round, every rank's group checked)."""
    import json

    port = _free_port()
    r = _run_bench_stub_lib(["--gpus", "8", "--steps", "8", "--warmup", "4"],
                            launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                      "8", "--master-addr", "127.0.0.1", "--master-port", str(port)], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    rec = json.loads(lines[0])
    _check_contract_line(rec, 8, 8)
    assert rec["extra"]["self_launched"] is False
    for blk, w in (("sharded_config4", 64), ("sharded_config5", 163)):
        b = rec[blk]
        assert "error" not in b, b
        assert b["group_size"] == 8 and b["n_groups"] == 1 and b["n_ranks_seen_by_rccl"] == 8
        assert set(b["variants"]) == {"replicated", "localq"} and b["all_variants_same_proof"] is True
        for v in b["variants"].values():
            assert v["all_ranks_same_proof"] is True and len(v["shard_stages_ms_per_rank"]) == 8
            assert v["collectives_count"] >= 1 and len(v["collectives_ms_total_per_rank"]) == 8
    assert rec["wall_budget"]["elapsed_total_s"] < rec["wall_budget"]["limit_s"]


def test_dry_run_scale_n1_equals_bench_n1():
    """The driver's SCALE series starts at N = 1 with the same command as BENCH: the same line (metric,
    config, the rank-0 legs with roofline and cpu_baseline keys) whether launched plainly or as a
    one-rank torch.distributed.run job."""
    import json

    plain = _run_bench_stub_lib(["--gpus", "1", "--steps", "8", "--warmup", "4", "--no-cpu-baseline"])
    assert plain.returncode == 0, plain.stderr[-3000:]
    a = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    port = _free_port()
    tr = _run_bench_stub_lib(["--gpus", "1", "--steps", "8", "--warmup", "4", "--no-cpu-baseline"],
                             launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                       "1", "--master-addr", "127.0.0.1", "--master-port", str(port)])
    assert tr.returncode == 0, tr.stderr[-3000:]
    b = json.loads([l for l in tr.stdout.splitlines() if l.startswith("{")][-1])
    for rec in (a, b):
        _check_contract_line(rec, 1, 8)
        assert rec["roofline"]["bound"] == "hbm" and "frac" in rec["roofline"] and "traffic" in rec["roofline"]
        assert "cpu_baseline" in rec and "stages_ms" in rec and rec["single_proof_latency_ms"] > 0
    strip = lambda c: {k: v for k, v in c.items() if k != "parallelism"}  # (it quotes the measured start spacing)
    assert a["metric"] == b["metric"] and strip(a["config"]) == strip(b["config"]) and set(a) == set(b)
