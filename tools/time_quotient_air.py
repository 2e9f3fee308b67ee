"""Quotient stage (uni-stark/src/prover.rs:122-194) on AIRs of growing size, through each of the three
device paths: the hiprtc-specialised kernel, the interpreter with its register file in LDS, and the
interpreter with the register file in a global slab (what a program with more live values than LDS holds
gets).  Kernel time from the library's per-kernel HIP events; one JSON line per (AIR, path).

    python tools/time_quotient_air.py [log_n=16] > gpurun_out/quotient_air.jsonl
"""
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd.airs import RandomAir, SynthExtAir, SynthMulAir, splitmix64_stream  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wait_big = os.environ.get("TS_TQ_WAIT_BIG", "1") != "0"
ctx = ts.default_context()
AIRS = [("SynthMulAir-64", SynthMulAir(64), 0), ("SynthExt-163", SynthExtAir(163), 0),
        ("Random(200 cols, 300 constraints, deg 5)", RandomAir(4242, 200, 300, 5, n_public=4, share_pct=35, max_depth=6), 4),
        ("Random(200 cols, 1000 constraints, deg 5)", RandomAir(4242, 200, 1000, 5, n_public=4, share_pct=35, max_depth=7), 4),
        ("Random(200 cols, 3000 constraints, deg 5)", RandomAir(4242, 200, 3000, 5, n_public=4, share_pct=20, max_depth=7), 4)]
only = os.environ.get("TS_TQ_ONLY")  # substring filter on the AIR name
paths = os.environ.get("TS_TQ_PATHS", "jit,interp-lds,interp-global").split(",")
n = 1 << log_n
for name, air, npub in AIRS:
    if only and only not in name:
        continue
    tape = ts.air_tape(air, npub)
    w = air.width()
    trace = splitmix64_stream(7, n * w).reshape(n, w)
    pis = splitmix64_stream(8, max(npub, 1))[:npub]
    alpha = splitmix64_stream(9, 4)
    ref = None
    for path in paths:
        env = {"jit": {}, "interp-lds": {"TS_NO_JIT": "1", "TS_INTERP_LDS_MAX_REGS": "1000000"},
               "interp-global": {"TS_NO_JIT": "1", "TS_INTERP_GLOBAL_REGS": "1"}, "interp": {"TS_NO_JIT": "1"}}[path]
        for k in ("TS_NO_JIT", "TS_INTERP_GLOBAL_REGS", "TS_INTERP_LDS_MAX_REGS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        t0 = time.time()
        cair = ts.CompiledAir(ctx, tape)
        prog = cair.program()
        rec = {"air": name, "path": path, "nodes": int(tape[4]), "constraints": int(tape[5]), "n_regs": prog["n_regs"],
               "n_instr": len(prog["code"]), "log_n": log_n}
        if path == "jit":
            if not cair.is_jit:
                if not wait_big and len(prog["code"]) > 6000:
                    rec["skipped"] = "background compilation not waited for"
                    print(json.dumps(rec), flush=True)
                    continue
                st, secs = cair.jit_wait()
                rec["background_compile_s"] = round(secs, 1)
                assert st == 3
            else:
                rec["sync_compile_s"] = round(time.time() - t0, 2)
        b = max(cair.log_quotient_degree, 1)
        pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 3, 2), ctx)
        _, data = pcs.commit([((log_n, 1), trace.copy())])
        chunks = pcs.quotient_chunks(data, cair, pis, alpha)  # warm (selector table)
        got = np.stack([c.download() for c in chunks])
        if ref is None:
            ref = got
        assert (got == ref).all(), "paths disagree"
        del chunks
        ctx.set_kernel_timing(True)
        reps = 3
        for _ in range(reps):
            pcs.quotient_chunks(data, cair, pis, alpha)
        kt = ctx.take_kernel_timings()
        ctx.set_kernel_timing(False)
        ms = sum(v[1] for k, v in kt.items() if "k_quotient" in k) / reps
        rows = n << cair.log_quotient_degree
        rec.update({"quotient_rows": rows, "kernel_ms": round(ms, 4),
                    "program_instructions_per_s": round(rows * len(prog["code"]) / (ms * 1e-3), 3),
                    "kernel": [k for k in kt if "k_quotient" in k][0],
                    "waves_per_cu": os.environ.get("TS_INTERP_WAVES_PER_CU")})
        print(json.dumps(rec), flush=True)
        os.environ.pop("TS_NO_JIT", None)
        os.environ.pop("TS_INTERP_GLOBAL_REGS", None)
