set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_parity.py -x -q -k "test_commit_lde_and_merkle" > gpurun_out/r3/t_lde.log 2>&1 || { tail -30 gpurun_out/r3/t_lde.log; exit 1; }
tail -3 gpurun_out/r3/t_lde.log
python3 -m pytest tests/test_gpu_sharded.py -x -q -k "config4" > gpurun_out/r3/t_c4.log 2>&1 || { tail -30 gpurun_out/r3/t_c4.log; exit 1; }
tail -3 gpurun_out/r3/t_c4.log
for lm in 12 0; do
  TS_LDE_LM=$lm python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > gpurun_out/r3/c4_lm$lm.json 2>> gpurun_out/r3/c4.err
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/c4_lm$lm.json')); print('LM', '$lm', d['ms_per_step'], d['single_proof_latency_ms'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof'])[:8]: print('   ', k, v['ms_per_proof'], v['avg_launch_ms'])"
done
