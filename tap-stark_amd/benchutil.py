"""Timing / aggregation protocol shared by bench.py and its CPU (gloo) test.

One rank per GPU.  Each rank times EXACTLY `steps` calls of its own `step()` between two
barriers; the job time is the MAX over ranks; the whole-job value is (units of all ranks) / time.
The proofs of different ranks are independent ("replicas"), so there is no
data-path collective: the only communication is the barrier and the max-reduction of the time.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Callable, Optional


@dataclass
class DistEnv:
    rank: int
    local_rank: int
    world: int
    dist: Optional[object] = None  # torch.distributed module when world > 1
    device: Optional[str] = None

    def barrier(self, local_sync: Callable[[], None]):
        local_sync()
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        import torch

        t = torch.tensor([value], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        import torch

        t = torch.tensor([value], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def init_dist(backend: Optional[str] = None) -> DistEnv:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (set by torch.distributed.run)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return DistEnv(rank, local_rank, 1)
    import torch
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # TS_BENCH_BACKEND=gloo rehearses the N > 1 path on boxes where RCCL cannot be used (e.g. all
    # ranks sharing one GPU); the default on GPUs is nccl (= RCCL on ROCm)
    backend = backend or os.environ.get("TS_BENCH_BACKEND") or (
        "nccl" if torch.cuda.is_available() else "gloo")
    device = None
    if backend == "nccl":
        if os.environ.get("TS_BENCH_SHARE_GPU"):
            # RCCL refuses two ranks on one device ("Duplicate GPU detected"): the shared-GPU rehearsal
            # covers the launch protocol and the host-staged (gloo) collectives only
            raise SystemExit("TS_BENCH_SHARE_GPU=1 needs TS_BENCH_BACKEND=gloo: RCCL cannot put two "
                             "ranks on one GPU")
        dev = local_rank
        torch.cuda.set_device(dev)
        device = f"cuda:{dev}"
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    return DistEnv(rank, local_rank, world, dist, device)


def run_timed(env: DistEnv, step: Callable[[int], None], steps: int, warmup: int,
              local_sync: Callable[[], None], units_per_step: float,
              run_steps: Optional[Callable[[int, int], None]] = None, extra_windows: int = 0) -> dict:
    """W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + device sync on
    both sides; returns the job-level numbers (identical on every rank).  `run_steps(first, count)`
    may replace the sequential loop (e.g. to keep several independent steps in flight); it must
    complete exactly the steps first .. first+count-1.  `extra_windows` more windows of K steps each
    follow the first, every one bracketed the same way: their ms/step are returned beside the first
    window's (`windows_ms_per_step`) so that the spread between windows is on the record; `value`,
    `ms_per_step` and `elapsed_s` are ALWAYS the first window's."""
    if run_steps is None:
        def run_steps(first, count):
            for i in range(first, first + count):
                step(i)
    run_steps(0, warmup)
    windows, marks = [], []
    for k in range(1 + max(0, extra_windows)):
        env.barrier(local_sync)
        t0 = time.perf_counter()
        if os.environ.get("TS_POOL_DEBUG"):  # same clock as the library's pool trace (CLOCK_MONOTONIC)
            import sys
            print(f"[bench t={time.monotonic() * 1e3:.3f} ms] timed window {k} starts", file=sys.stderr, flush=True)
        run_steps(warmup + k * steps, steps)
        env.barrier(local_sync)
        t1 = time.perf_counter()
        marks.append((t0, t1))  # this rank's clock (time.perf_counter), for per-window sampler / lane statistics
        windows.append(env.max_over_ranks(t1 - t0))
    elapsed = windows[0]
    total_units = env.sum_over_ranks(units_per_step * steps)
    return {"elapsed_s": elapsed, "ms_per_step": 1e3 * elapsed / steps,
            "value": total_units / elapsed, "steps_per_sec": env.world * steps / elapsed,
            "windows_ms_per_step": [1e3 * e / steps for e in windows], "windows_t": marks}


def split_stage_timings(items) -> tuple[dict, list]:
    """``Context.take_timings()`` of a sharded proof -> (stage name -> summed ms, collectives table).
    Every collective leaves an entry "collective: <kind> <bytes> B... (<site>)" (csrc/sharded.cpp);
    the table has one row per (kind, bytes, site): count, total and largest ms."""
    stages, coll = {}, {}
    for k, v in items:
        if k.startswith("collective: "):
            row = coll.setdefault(k[len("collective: "):], {"count": 0, "ms_total": 0.0, "ms_max": 0.0})
            row["count"] += 1
            row["ms_total"] += v
            row["ms_max"] = max(row["ms_max"], v)
        else:
            stages[k] = round(stages.get(k, 0.0) + v, 3)
    table = []
    for k, row in coll.items():
        kind, rest = k.split(" ", 1)
        nbytes = int(rest.split(" ", 1)[0])
        table.append({"collective": kind, "what": k, "bytes": nbytes, "count": row["count"],
                      "ms_total": round(row["ms_total"], 3), "ms_max": round(row["ms_max"], 3)})
    return stages, table


_SAMPLER_SRC = r"""
import json, select, sys, time
import amdsmi
amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[int(sys.argv[1])]
period = float(sys.argv[2])
out = open(sys.argv[3], "w")  # a file, not the pipe: a long run must not block on a full pipe
sys.stdout.write("ready\n"); sys.stdout.flush()
while True:
    if select.select([sys.stdin], [], [], 0)[0]:
        break  # the parent closed / wrote to our stdin: stop
    t = time.perf_counter()  # CLOCK_MONOTONIC: the same clock as the parent's perf_counter
    try:
        m = amdsmi.amdsmi_get_gpu_metrics_info(h)
        clks = [c for c in m.get("current_gfxclks", []) if isinstance(c, int) and 0 < c < 10000]
        out.write(json.dumps({"t": t, "clk": (sum(clks) / len(clks)) if clks else m.get("current_gfxclk"),
                              "w": m.get("current_socket_power"), "uclk": m.get("current_uclk"),
                              "temp": m.get("temperature_hotspot")}) + "\n")
    except Exception as e:
        out.write(json.dumps({"t": t, "error": repr(e)}) + "\n")
    rest = period - (time.perf_counter() - t)
    if rest > 0:
        time.sleep(rest)
out.flush()
"""


class GpuSamplerProcess:
    """The same samples from a CHILD process (amdsmi only, no HIP): usable INSIDE the timed region,
    because it takes no time from this process's interpreter lock -- a sampler thread here would hold
    it for a fraction of a millisecond per sample, and the lanes need it between two proofs.  The child
    stamps samples with time.perf_counter (CLOCK_MONOTONIC, shared by all processes of the machine), so
    `window(t0, t1)` can cut them by this process's own timestamps."""

    def __init__(self, device_index: int = 0, period_s: float = 0.01):
        import subprocess
        import sys
        import tempfile
        self.samples, self.error, self._p = [], None, None
        fd, self._path = tempfile.mkstemp(prefix="ts_smi_", suffix=".jsonl")
        os.close(fd)
        try:
            self._p = subprocess.Popen([sys.executable, "-c", _SAMPLER_SRC, str(device_index), str(period_s), self._path],
                                       stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                       text=True)
            first = self._p.stdout.readline()
            if first.strip() != "ready":
                self.error = "sampler child did not start (amdsmi missing?)"
                self._p.kill()
                self._p = None
        except Exception as e:  # noqa: BLE001
            self.error = repr(e)
            self._p = None

    def stop(self):
        import json
        if self._p is None:
            return
        out = ""
        try:
            self._p.stdin.close()
            self._p.wait(timeout=5)
        except Exception as e:  # noqa: BLE001
            self.error = repr(e)
            self._p.kill()
        self._p = None
        try:
            with open(self._path) as f:
                out = f.read()
            os.unlink(self._path)
        except OSError as e:
            self.error = repr(e)
        for line in out.splitlines():
            try:
                s = json.loads(line)
            except ValueError:
                continue
            if "error" in s:
                self.error = s["error"]
            else:
                self.samples.append(s)

    def window(self, t0: float, t1: float) -> dict:
        def med(xs):
            xs = sorted(x for x in xs if isinstance(x, (int, float)))
            return xs[len(xs) // 2] if xs else None
        ss = [s for s in self.samples if t0 <= s["t"] <= t1]
        clk = [s["clk"] for s in ss]
        return {"samples": len(ss), "gfxclk_mhz_median": med(clk),
                "gfxclk_mhz_min": min((c for c in clk if isinstance(c, (int, float))), default=None),
                "socket_power_w_median": med([s["w"] for s in ss]),
                "hotspot_c_max": max((s["temp"] for s in ss if isinstance(s["temp"], (int, float))), default=None)}


class GpuSampler:
    """Samples shader clock (per XCD), socket power and hotspot temperature of one GPU from a thread
    (amdsmi, readable by an ordinary user on the GPU box; ~0.5 ms per sample).  Used OUTSIDE the timed
    region: a sustained run of the same workload is sampled so that the clock the chip holds under
    this load is a measurement on the record, not an inference from SQ_BUSY_CYCLES."""

    def __init__(self, device_index: int = 0, interval_s: float = 0.01):
        self.interval = interval_s
        self.samples = []
        self.error = None
        self._stop = None
        self._thread = None
        try:
            import amdsmi
            self._smi = amdsmi
            amdsmi.amdsmi_init()
            self._h = amdsmi.amdsmi_get_processor_handles()[device_index]
        except Exception as e:  # noqa: BLE001 -- the sampler is optional everywhere
            self._smi = None
            self.error = repr(e)

    def read(self):
        m = self._smi.amdsmi_get_gpu_metrics_info(self._h)
        clks = [c for c in m.get("current_gfxclks", []) if isinstance(c, int) and 0 < c < 10000]
        return {"t": time.perf_counter(), "gfxclk_mhz": clks or [m.get("current_gfxclk")],
                "socket_power_w": m.get("current_socket_power"), "hotspot_c": m.get("temperature_hotspot"),
                "uclk_mhz": m.get("current_uclk")}

    def __enter__(self):
        import threading
        self.samples = []
        if self._smi is None:
            return self
        self._stop = threading.Event()

        def run():
            while not self._stop.is_set():
                try:
                    self.samples.append(self.read())
                except Exception as e:  # noqa: BLE001
                    self.error = repr(e)
                    return
                self._stop.wait(self.interval)

        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *a):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2)

    def summary(self) -> dict:
        if not self.samples:
            return {"error": self.error or "no samples"}

        def med(xs):
            xs = sorted(x for x in xs if isinstance(x, (int, float)))
            return xs[len(xs) // 2] if xs else None
        per = [sum(s["gfxclk_mhz"]) / len(s["gfxclk_mhz"]) for s in self.samples if s["gfxclk_mhz"] and s["gfxclk_mhz"][0]]
        return {"samples": len(self.samples), "gfxclk_mhz_median": med(per),
                "gfxclk_mhz_min": min(per) if per else None, "gfxclk_mhz_max": max(per) if per else None,
                "socket_power_w_median": med([s["socket_power_w"] for s in self.samples]),
                "socket_power_w_max": max([s["socket_power_w"] for s in self.samples
                                           if isinstance(s["socket_power_w"], (int, float))], default=None),
                "hotspot_c_max": max([s["hotspot_c"] for s in self.samples
                                      if isinstance(s["hotspot_c"], (int, float))], default=None),
                "uclk_mhz_median": med([s["uclk_mhz"] for s in self.samples])}
