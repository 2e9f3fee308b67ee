#!/bin/bash
# headline (config 3) against the number of proofs in flight and the start gate
O=gpurun_out/r5
mkdir -p $O
for s in 3 4 5 6 8; do
  python bench.py --streams $s --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $s', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['config']['parallelism'][-40:])"
done > $O/streams.txt 2>&1
for g in 0 0.5 1.5; do
  python bench.py --streams 4 --stagger-ms $g --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams 4 stagger $g', d['ms_per_step'], d['extra']['windows_ms_per_step'])"
done >> $O/streams.txt 2>&1
cat $O/streams.txt
