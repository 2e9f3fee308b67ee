"""What does LDS traffic cost in power?  The butterfly loop of ts_bench_alu with and without the LDS round
trips of a contiguous NTT pass (one 4-byte LDS access per butterfly), each in a sustained loop sampled
with amdsmi (clock, socket power).  Kinds: 0 registers only; 3 + ds_write_b32 / ds_read_b32 on the padded
image; 4 the same bytes as 16-byte accesses; 1 Blake3 for reference.

    python tools/power_alu_lds.py > profiles/r05_power_alu_lds.json
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
out = {"_comment": __doc__.split("\n\n")[0]}
with smp:
    time.sleep(0.3)
out["idle"] = smp.summary()
for kind, name in ((0, "butterflies, registers only"), (3, "butterflies + LDS round trip per radix-16 round (b32)"),
                   (4, "butterflies + the same bytes as b128 accesses"), (1, "blake3 compressions"), (0, "butterflies again")):
    ctx.alu_ceiling(kind)
    rates = []
    with smp:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.8:
            rates.append(ctx.alu_ceiling(kind))
    s = smp.summary()
    r = dict(kind=kind, name=name, units_per_s=sorted(rates)[len(rates) // 2], gfxclk_mhz_median=s.get("gfxclk_mhz_median"),
             socket_power_w_median=s.get("socket_power_w_median"), socket_power_w_max=s.get("socket_power_w_max"))
    idle = out["idle"].get("socket_power_w_median") or 0
    r["nj_per_unit_total"] = round(r["socket_power_w_median"] / r["units_per_s"] * 1e9, 4)
    r["nj_per_unit_above_idle"] = round((r["socket_power_w_median"] - idle) / r["units_per_s"] * 1e9, 4)
    out.setdefault("rows", []).append(r)
    print(r, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
