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
