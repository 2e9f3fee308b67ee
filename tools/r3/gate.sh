set -e
mkdir -p gpurun_out/r3
for st in -1 0.7 1.0 1.4 2.0 -1 1.0 1.4; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --headline-only --windows 5 --stagger-ms $st > gpurun_out/r3/g_$st.json 2>> gpurun_out/r3/g.err
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r3/g_$st.json')); print('gate', '$st', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['extra']['stagger_ms'])"
done
