set -e
mkdir -p gpurun_out/r3
for g in 0 1 0 1; do TS_FRI_GRAPH=$g python3 tools/latency.py config3 config2 2>> gpurun_out/r3/graph.err | tee -a gpurun_out/r3/graph.log; done
for g in 0 1; do
TS_FRI_GRAPH=$g python3 bench.py --steps 20 --warmup 5 --headline-only --windows 3 > gpurun_out/r3/graph_b$g.json 2>> gpurun_out/r3/graph.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/graph_b$g.json')); print('throughput TS_FRI_GRAPH=$g', d['ms_per_step'], d['extra']['windows_ms_per_step'])" | tee -a gpurun_out/r3/graph.log
done
