set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_mmcs.py tests/test_gpu_parity.py tests/test_gpu_random_shapes.py -x -q > gpurun_out/r3/t_leaf.log 2>&1 || { tail -30 gpurun_out/r3/t_leaf.log; exit 1; }
tail -2 gpurun_out/r3/t_leaf.log
for cfg in config5:2:8 config2:8:48; do
  IFS=: read w s k <<< "$cfg"
  python3 bench.py --workload $w --streams $s --steps $k --warmup $s --windows 2 --no-cpu-baseline > gpurun_out/r3/leaf_$w.json 2>> gpurun_out/r3/ab.err
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/leaf_$w.json')); print('$w', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['single_proof_latency_ms'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof']):
    if 'leaf' in k: print('   ', k, v['launches_per_proof'], v['ms_per_proof'])"
done
