"""Which part of the prover is power-limited?  Sustained loops of single stages on ONE context, each
sampled with amdsmi (clock per XCD, socket power) for ~0.6 s:

  alu butterflies / alu blake3   register-resident loops (ts_bench_alu)
  merkle commit                  Blake3 leaf hashes + levels of a resident 2^22 x 64 matrix (ts_mmcs_commit)
  pcs commit                     the same + transpose + the three NTT passes (ts_pcs_commit on 2^20 x 64)
  whole proofs                   ts_prove, one lane

    python tools/power_per_stage.py > profiles/r04_power_per_stage.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import tapstark_amd as ts  # noqa: E402
from tapstark_amd.airs import SynthMulAir  # noqa: E402
from tapstark_amd.benchutil import GpuSampler  # noqa: E402

ctx = ts.default_context()
smp = GpuSampler(0, 0.01)
out = {"_comment": __doc__.split("\n\n")[0]}
n, w = 1 << 20, 64


def sampled(name, fn, seconds=0.6):
    fn()
    ctx.synchronize()
    with smp:
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < seconds:
            fn()
            k += 1
        ctx.synchronize()
        dt = time.perf_counter() - t0
    out[name] = dict(smp.summary(), calls=k, ms_per_call=round(1e3 * dt / k, 4))


with smp:
    time.sleep(0.3)
out["idle"] = smp.summary()
sampled("alu butterflies", lambda: ctx.alu_ceiling(0))
sampled("alu blake3", lambda: ctx.alu_ceiling(1))
mmcs = ts.Blake3Mmcs(ctx)
import torch  # noqa: E402  (resident sources: commit consumes its matrix, so every call takes a device-to-device copy)

src_big = torch.randint(0, 0x78000001, (4 * n, w), dtype=torch.int32, device="cuda:0")
src = src_big[:n].contiguous()
torch.cuda.synchronize()
sampled("device-to-device copy of 2^22 x 64 (the source of the next row)",
        lambda: ts.DeviceMatrix.from_device_ptr(ctx, src_big.data_ptr(), 4 * n, w))
sampled("merkle commit (copy + leaf hash + levels, 2^22 x 64)",
        lambda: mmcs.commit([ts.DeviceMatrix.from_device_ptr(ctx, src_big.data_ptr(), 4 * n, w)]))
pcs = ts.TwoAdicFriPcs(ts.FriConfig(2, 28, 8), ctx)
sampled("pcs commit (copy + transpose + LDE + merkle, 2^20 x 64, log_blowup 2)",
        lambda: pcs.commit([((20, 1), ts.DeviceMatrix.from_device_ptr(ctx, src.data_ptr(), n, w))]))
config = ts.StarkConfig(pcs)
cair = ts.CompiledAir(ctx, ts.air_tape(SynthMulAir(64), 0))
pis = np.zeros(0, dtype=np.uint32)
sampled("whole proofs, one lane", lambda: ts.prove(config, cair, ts.BfChallenger(), ts.DeviceMatrix.synth_mul(ctx, n, w), pis))
print(json.dumps(out, indent=1))
