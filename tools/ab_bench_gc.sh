#!/bin/bash
# Hunting the sporadic slow window: many short headline runs, printing for every window its ms/step and, for the
# slow ones, every proof's wall time lane by lane.   bash tools/ab_bench_gc.sh  ->  gpurun_out/r6/window_hunt.txt
mkdir -p gpurun_out/r6
for i in $(seq 1 ${1:-10}); do
  for S in 1 0; do
  TS_BENCH_SAMPLER=$S python bench.py --headline-only --windows 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
w=d['windows_ms_per_step']; print('sampler=$S', w, d['priming']['probes_ms_per_step'])
med=sorted(w)[len(w)//2]
for r in d['windows']:
    if r['ms_per_step'] > 1.04*med: print('   SLOW window', r['window'], r['ms_per_step'], 'clk', r.get('gfxclk_mhz_median'), 'gap', r.get('longest_host_gap_ms'), r.get('proof_latencies_ms_by_lane'))
" >> gpurun_out/r6/window_hunt.txt
  done
done
cat gpurun_out/r6/window_hunt.txt
