#!/usr/bin/env python3
"""The reference's only benchmark on this hardware: fri/benches/fold_even_odd.rs:14-46 sweeps a fold
over vectors of 2^12 .. 2^22 elements.  Here: FriGenericConfig::fold_matrix (two_adic_pcs.rs:116-147,
the fold the prover runs) on EF4 vectors of those sizes, device-resident (ts_fri_fold_device ->
k_fri_fold_pairs), timed with HIP events around every launch on the library's stream; the oracle's
fold_matrix on the host beside it.  Bytes per fold: 2h EF4 read + h EF4 written + h twiddles = 52 h.

    python tools/bench_fold.py > profiles/r03_fold_even_odd.json
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402  (host baseline + result check)
from tapstark_amd import _lib  # noqa: E402

P = 0x78000001


def main():
    ctx = ts.default_context()
    l = _lib.lib()
    rng = np.random.default_rng(3)
    beta = rng.integers(0, P, 4, dtype=np.uint32)
    rows = []
    for log_size in (12, 14, 16, 18, 20, 22):  # the reference's sweep: elements of the INPUT vector
        n = 1 << log_size
        h = n // 2
        vec = rng.integers(0, P, (n, 4), dtype=np.uint32)
        d_in = torch.from_numpy(vec.view(np.int32)).to("cuda:0")
        d_out = torch.zeros((h, 4), dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        bp = beta.ctypes.data_as(C.POINTER(C.c_uint32))

        def fold():
            ctx.check(l.ts_fri_fold_device(ctx.h, d_in.data_ptr(), h, bp, d_out.data_ptr()))
        fold()
        ctx.synchronize()
        got = d_out.cpu().numpy().view(np.uint32)
        want = orc.fold_matrix(vec, beta)
        assert (got == want).all(), f"fold of 2^{log_size} differs from the oracle"
        reps = 50
        ctx.set_kernel_timing(True)
        for _ in range(reps):
            fold()
        kt = ctx.take_kernel_timings()
        ctx.set_kernel_timing(False)
        name = next(k for k in kt if "fold" in k)
        us = 1e3 * kt[name][1] / kt[name][0]
        # wall per call from the host (what a criterion-style loop around the call would see)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fold()
        ctx.synchronize()
        wall_us = 1e6 * (time.perf_counter() - t0) / reps
        # host: the oracle's fold_matrix (C, one thread), best of 5
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            orc.fold_matrix(vec, beta)
            best = min(best, time.perf_counter() - t0)
        nbytes = 52 * h
        rows.append({"log_size": log_size, "elements_in": n, "kernel": name, "kernel_us": round(us, 3),
                     "call_us_host_wall": round(wall_us, 3), "alg_bytes": nbytes,
                     "GB_per_s": round(nbytes / (us * 1e-6) / 1e9, 1),
                     "frac_of_8TBps": round(nbytes / (us * 1e-6) / 8e12, 4),
                     "oracle_host_us": round(1e6 * best, 1),
                     "elements_per_s_gpu": round(n / (us * 1e-6)), "elements_per_s_host": round(n / best)})
    print(json.dumps({"benchmark": "fold_matrix on EF4 vectors, sizes of fri/benches/fold_even_odd.rs:14-46",
                      "bytes_per_fold": "52 h (2h EF4 in, h EF4 out, h twiddles)",
                      "note": "below ~2^18 elements a launch is latency-bound (the kernel floor as HIP events see "
                              "it is ~2-7 us); the prover itself never launches such folds alone: rounds with <= 2^17 "
                              "leaves fold inside k_fri_round / k_fri_tail",
                      "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
