"""One rank of a sharded proof (launched by test_gpu_sharded.py; not a test module itself).

argv[1] = JSON spec {air, log_n, cfg, world, backend, port, min_local_log, out}.  Rank r proves with
the row slice r of the trace and writes the proof words to <out>.rank<r>.npy.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_case(name, log_n):
    from tapstark_amd.airs import (FibonacciAir, SynthExtAir, SynthMulAir, fibonacci_public_values,
                                   generate_fibonacci_trace, generate_synth_ext_trace,
                                   generate_synth_mul_trace)

    n = 1 << log_n
    if name == "fib":
        tr = generate_fibonacci_trace(0, 1, n)
        return FibonacciAir(), tr, fibonacci_public_values(tr)
    if name.startswith("mul"):
        w = int(name[3:])
        return SynthMulAir(w), generate_synth_mul_trace(n, w), np.zeros(0, dtype=np.uint32)
    if name.startswith("ext"):
        w = int(name[3:])
        return SynthExtAir(w), generate_synth_ext_trace(n, w), np.zeros(0, dtype=np.uint32)
    raise ValueError(name)


def main():
    spec = json.loads(sys.argv[1])
    rank = int(os.environ["RANK"])
    world = spec["world"]
    import torch
    import torch.distributed as dist

    import tapstark_amd as ts
    from tapstark_amd.dist import TorchComm

    torch.cuda.set_device(0)
    dist.init_process_group(backend=spec["backend"], init_method=f"tcp://127.0.0.1:{spec['port']}",
                            rank=rank, world_size=world)
    ctx = ts.default_context()
    air, trace, pis = make_case(spec["air"], spec["log_n"])
    n = trace.shape[0]
    if spec.get("replicated"):
        rows = trace  # every rank holds the whole trace: no all-gather of it
    else:
        rows = np.ascontiguousarray(trace[rank * n // world:(rank + 1) * n // world])
    config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(*spec["cfg"]), ctx))
    comm = TorchComm(0)
    challenger = ts.BfChallenger()
    proof = ts.prove_sharded(config, air, challenger, rows, pis, comm, spec["min_local_log"],
                             trace_replicated=bool(spec.get("replicated")))
    np.save(f"{spec['out']}.rank{rank}.npy", proof.words)
    with open(f"{spec['out']}.rank{rank}.json", "w") as f:
        json.dump({"calls": comm.calls, "chal_bits": challenger.sample_bits(20)}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
