"""Per-kernel HIP-event times of whole C3 proofs (same-box A/B with TS_LIB_PATH): time_kernels.py [name-substring]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tapstark_amd as ts
from tapstark_amd.airs import SynthMulAir

pat = sys.argv[1] if len(sys.argv) > 1 else ""
ctx = ts.default_context()
config = ts.StarkConfig(ts.TwoAdicFriPcs(ts.FriConfig(2, 28, 8), ctx))
cair = ts.CompiledAir(ctx, ts.air_tape(SynthMulAir(64), 0))
ts.prove(config, cair, ts.BfChallenger(), ts.DeviceMatrix.synth_mul(ctx, 1 << 20, 64), [])
ctx.set_kernel_timing(True)
reps = 4
for _ in range(reps):
    ts.prove(config, cair, ts.BfChallenger(), ts.DeviceMatrix.synth_mul(ctx, 1 << 20, 64), [])
kt = ctx.take_kernel_timings()
for k, (cnt, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1]):
    if pat in k:
        print(f"{k:40s} {cnt / reps:6.1f} launches  {ms / reps:8.4f} ms/proof")
