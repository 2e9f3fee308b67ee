#!/bin/bash
# Same-box A/B of the host's stream wait: sleeping on the interrupt (hipStreamSynchronize) against polling
# (TS_SYNC_SPIN=1): single-proof latency and the 4-lane windows, processes alternating.
#     bash tools/ab_sync_spin.sh [pairs=5]  ->  gpurun_out/r6/sync_spin_ab.txt
mkdir -p gpurun_out/r6
O=gpurun_out/r6/sync_spin_ab.txt
for i in $(seq 1 ${1:-5}); do
  for S in 0 1; do
    TS_SYNC_SPIN=$S python tools/latency_simple.py 40 2>/dev/null >> $O
    TS_SYNC_SPIN=$S python bench.py --headline-only --windows 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TS_SYNC_SPIN=$S  windows', d['windows_ms_per_step'], 'probes', d['priming']['probes_ms_per_step'])" >> $O
  done
done
cat $O
