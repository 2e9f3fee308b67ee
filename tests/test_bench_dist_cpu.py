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
