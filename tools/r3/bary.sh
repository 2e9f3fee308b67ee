set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_random_shapes.py -x -q -k "prove or open or eight_ranks or random" > gpurun_out/r3/t_bary.log 2>&1 || { tail -30 gpurun_out/r3/t_bary.log; exit 1; }
tail -1 gpurun_out/r3/t_bary.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/bary_c3.json 2>> gpurun_out/r3/ab.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/bary_c3.json')); print('C3', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['single_proof_latency_ms'], d['roofline']['kernel_ms_total_per_proof'], d['stages_ms']['compute opened values with Lagrange interpolation'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof']):
    if 'bary' in k: print('   ', k, v['launches_per_proof'], v['ms_per_proof'])"
python3 tools/latency.py config3 config2 2>/dev/null
