#!/bin/bash
# The sporadic slow window: does it belong to the HIP runtime torch's wheel bundles (the one a Python process that
# imported torch is bound to) or to the system's?   bash tools/ab_bench_hip.sh [pairs=12]  ->  gpurun_out/r6/hip_ab.txt
mkdir -p gpurun_out/r6
for i in $(seq 1 ${1:-12}); do
  for NT in 1 0; do
  TS_BENCH_NO_TORCH=$NT python bench.py --headline-only --windows 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no_torch=$NT', d['windows_ms_per_step'], d['priming']['probes_ms_per_step'], max(w.get('proof_latency_in_window_ms_max') or 0 for w in d['windows']))" >> gpurun_out/r6/hip_ab.txt
  done
done
cat gpurun_out/r6/hip_ab.txt
