"""Per-kernel HIP-event timings of Pcs::commit for one shape: time_commit.py LOG_N WIDTH LOG_BLOWUP [REPS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tapstark_amd as ts
from tapstark_amd.airs import splitmix64_stream

log_n, w, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = ts.default_context()
pcs = ts.TwoAdicFriPcs(ts.FriConfig(b, 16, 8), ctx)
m = splitmix64_stream(7, (1 << log_n) * w).reshape(1 << log_n, w)
root, data = pcs.commit([((log_n, 1), ts.DeviceMatrix.upload(ctx, m))])
del data
mats = [ts.DeviceMatrix.upload(ctx, m) for _ in range(reps)]
ctx.set_kernel_timing(True)
for dm in mats:
    root, data = pcs.commit([((log_n, 1), dm)])
    del data
kt = ctx.take_kernel_timings()
n, N = 1 << log_n, 1 << (log_n + b)
for k, (cnt, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:36s} {cnt/reps:5.1f} launches  {ms/reps:8.3f} ms/commit")
print("root", root[:2], "LDE GB", 4 * N * w / 1e9)
