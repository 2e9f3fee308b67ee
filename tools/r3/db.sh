set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded.py -x -q -k "test_commit_lde_and_merkle or prove or config4 or eight_ranks" > gpurun_out/r3/t_db.log 2>&1 || { tail -30 gpurun_out/r3/t_db.log; exit 1; }
tail -2 gpurun_out/r3/t_db.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/db_c3.json 2>> gpurun_out/r3/ab.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/db_c3.json')); print('C3', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['single_proof_latency_ms'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof'])[:3]: print('   ', k, v['ms_per_proof'])"
python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > gpurun_out/r3/db_c4.json 2>> gpurun_out/r3/ab.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/db_c4.json')); print('C4', d['ms_per_step'], d['single_proof_latency_ms'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof'])[:3]: print('   ', k, v['ms_per_proof'])"
