set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_parity.py -x -q -k "test_commit_lde_and_merkle" > gpurun_out/r3/t_occ.log 2>&1 || { tail -30 gpurun_out/r3/t_occ.log; exit 1; }
tail -1 gpurun_out/r3/t_occ.log
for i in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3/occ_c3.json 2>> gpurun_out/r3/ab.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/occ_c3.json')); print('C3', d['ms_per_step'], d['extra']['windows_ms_per_step'], d['single_proof_latency_ms'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof'])[:2]: print('   ', k, v['ms_per_proof'])"
done
python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > gpurun_out/r3/occ_c4.json 2>> gpurun_out/r3/ab.err
python3 -c "
import json
d=json.load(open('gpurun_out/r3/occ_c4.json')); print('C4', d['ms_per_step'])
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_per_proof'])[:2]: print('   ', k, v['ms_per_proof'])"
