#!/bin/bash
# Do the slow windows coincide with cgroup CPU throttling of the container?  Short headline runs; per run the windows'
# ms/step with the number of throttle events in each, then the same for the priming probes.
#     bash tools/ab_bench_throttle.sh [runs=16] -> gpurun_out/r6/throttle_hunt.txt
mkdir -p gpurun_out/r6
O=gpurun_out/r6/throttle_hunt.txt
echo "nproc $(nproc), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" >> $O
for i in $(seq 1 ${1:-16}); do
  python bench.py --headline-only --windows 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
w=[(r['ms_per_step'], (r.get('cgroup_cpu_throttled') or {}).get('times')) for r in d['windows']]
p=list(zip(d['priming']['probes_ms_per_step'], d['priming']['probes_cgroup_cpu_throttled_times']))
print('windows', w, ' probes', p)" >> $O
done
cat $O
