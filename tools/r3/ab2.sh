set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -k "test_commit_lde_and_merkle or prove or column_sharded or eight_ranks" > gpurun_out/r3/t_fused.log 2>&1 || { tail -30 gpurun_out/r3/t_fused.log; exit 1; }
tail -2 gpurun_out/r3/t_fused.log
for v in base fused base2 fused2; do
  case $v in base*) export TS_LDE_NO_FUSED_TRANSPOSE=1;; *) unset TS_LDE_NO_FUSED_TRANSPOSE;; esac
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/c3_$v.json 2>> gpurun_out/r3/ab.err
  python3 -c "
import json
d=json.load(open('gpurun_out/r3/c3_$v.json')); print('C3 $v', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['single_proof_latency_ms'], d['roofline']['kernel_ms_total_per_proof'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof']):
    if 'intt' in k or 'transpose' in k: print('   ', k, v['ms_per_proof'], v['avg_launch_ms'])"
done
