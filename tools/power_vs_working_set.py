"""Is HBM traffic what limits the NTT passes (through the power cap), or VALU issue?  

The SAME kernels -- the coset LDE of a 2^20-row matrix, log_blowup 2, and the Merkle hashing of the
result -- are run in sustained loops on matrices of 64, 32, 16 and 8 columns.  Per column the VALU work
is identical (the kernels are per column, grids stay >= 1024 workgroups); what changes is where the
bytes live: 64 columns = 1.34 GB per repetition (HBM), 8 columns = 168 MB (inside the 256 MB
Infinity Cache).  Each loop (~0.8 s) is sampled with amdsmi: shader clock per XCD, socket power.

    ms_per_column falling and the clock rising as the working set leaves HBM  ->  bytes are watts, worth attacking
    ms_per_column and clock flat                                              ->  VALU issue binds, traffic does not

    python tools/power_vs_working_set.py > profiles/r05_power_vs_working_set.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import tapstark_amd as ts  # noqa: E402
from tapstark_amd.benchutil import GpuSampler  # noqa: E402

ctx = ts.default_context()
smp = GpuSampler(0, 0.01)
out = {"_comment": __doc__.split("\n\n")[0], "log_n": 20, "log_blowup": 2, "rows": []}
LOG_N, B = 20, 2


def sustained(stage, width, seconds=0.8):
    one = ctx.bench_stage(stage, LOG_N, width, B, 2)
    reps = max(2, int(seconds * 1e3 / one))
    with smp:
        ms = ctx.bench_stage(stage, LOG_N, width, B, reps)
    s = smp.summary()
    n, N = 1 << LOG_N, 1 << (LOG_N + B)
    foot = 4 * width * (n + N) if stage == 0 else 4 * width * N + 64 * N
    return dict(stage=["coset_lde", "merkle_commit"][stage], width=width, reps=reps, ms_per_rep=round(ms, 4),
                us_per_column=round(1e3 * ms / width, 3), footprint_mb=round(foot / 1e6, 1),
                gfxclk_mhz_median=s.get("gfxclk_mhz_median"), gfxclk_mhz_min=s.get("gfxclk_mhz_min"),
                socket_power_w_median=s.get("socket_power_w_median"), socket_power_w_max=s.get("socket_power_w_max"),
                samples=s.get("samples"))


with smp:
    time.sleep(0.3)
out["idle"] = smp.summary()
with smp:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:
        ctx.alu_ceiling(0)
out["alu butterflies (register-resident loop)"] = smp.summary()
for stage in (0, 1):
    for width in (64, 32, 16, 8, 64):  # 64 again at the end: drift of the box over the run
        r = sustained(stage, width)
        out["rows"].append(r)
        print(r, file=sys.stderr, flush=True)
# The other way to shrink the working set: fewer ROWS at 64 columns.  Grids keep their width (every
# column is there), the passes change shape with n (17 .. 20 stages), so the unit is the butterfly.
out["rows_by_height"] = []
for log_n in (17, 18, 19, 20):
    LOG_N = log_n
    r = sustained(0, 64)
    bf = 5 * (1 << (log_n - 1)) * log_n * 64  # inverse + 4 cosets forward
    r.update(log_n=log_n, butterflies=bf, ps_per_butterfly=round(1e9 * r["ms_per_rep"] / bf, 4))
    out["rows_by_height"].append(r)
    print(r, file=sys.stderr, flush=True)
# Which of the three LDE passes sits at the cap?  Each pass alone (ts_bench_stage 2 / 3 / 4), 2^20 x 64.
LOG_N = 20
out["rows_by_pass"] = []
for st, nm in ((2, "k_intt_contig (inverse, contiguous stages)"), (3, "k_lde_mid (strided inverse + scale + strided forward, 4 cosets)"),
               (4, "k_lde_fwd_contig (forward, contiguous stages, 4 cosets)")):
    one = ctx.bench_stage(st, LOG_N, 64, B, 2)
    reps = max(2, int(800 / one))
    with smp:
        ms = ctx.bench_stage(st, LOG_N, 64, B, reps)
    s = smp.summary()
    r = dict(stage=nm, reps=reps, ms_per_rep=round(ms, 4), gfxclk_mhz_median=s.get("gfxclk_mhz_median"),
             socket_power_w_median=s.get("socket_power_w_median"), socket_power_w_max=s.get("socket_power_w_max"))
    out["rows_by_pass"].append(r)
    print(r, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
