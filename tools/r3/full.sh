set -e
mkdir -p gpurun_out/r3
python3 tools/bench_fold.py > gpurun_out/r3/fold.json 2> gpurun_out/r3/fold.err || { tail -20 gpurun_out/r3/fold.err; exit 1; }
python3 -c "
import json
for r in json.load(open('gpurun_out/r3/fold.json'))['rows']: print(r['log_size'], r['kernel_us'], r['GB_per_s'], r['frac_of_8TBps'], r['oracle_host_us'])"
python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/t_full.log 2>&1 || { tail -40 gpurun_out/r3/t_full.log; exit 1; }
tail -3 gpurun_out/r3/t_full.log
python3 __graft_entry__.py --smoke 2>&1 | tail -2
