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
        dev = 0 if os.environ.get("TS_BENCH_SHARE_GPU") else local_rank  # rehearsal: every rank on GPU 0
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
    windows = []
    for k in range(1 + max(0, extra_windows)):
        env.barrier(local_sync)
        t0 = time.perf_counter()
        if os.environ.get("TS_POOL_DEBUG"):  # same clock as the library's pool trace (CLOCK_MONOTONIC)
            import sys
            print(f"[bench t={time.monotonic() * 1e3:.3f} ms] timed window {k} starts", file=sys.stderr, flush=True)
        run_steps(warmup + k * steps, steps)
        env.barrier(local_sync)
        windows.append(env.max_over_ranks(time.perf_counter() - t0))
    elapsed = windows[0]
    total_units = env.sum_over_ranks(units_per_step * steps)
    return {"elapsed_s": elapsed, "ms_per_step": 1e3 * elapsed / steps,
            "value": total_units / elapsed, "steps_per_sec": env.world * steps / elapsed,
            "windows_ms_per_step": [1e3 * e / steps for e in windows]}
