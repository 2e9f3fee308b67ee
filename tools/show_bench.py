"""Print the headline fields of bench.py JSON logs (last line of each file)."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f"{f}: ms/step {d['ms_per_step']:.3f}  value {d['value']:.4g}  proofs/s {d['proofs_per_sec']:.1f}  "
              f"[{d['config']['parallelism']}]  dom {r['kernel']} frac {r['frac']}\n   stages {d['stages_ms']}")
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e, open(f).read()[-1500:])


def kernels(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_proof"]):
        print(f"  {k:40s} {v['launches_per_proof']:6.1f} {v['ms_per_proof']:8.3f} ms  {v['alg_gbps']}")


if len(sys.argv) > 1 and sys.argv[-1] == "-k":
    pass
