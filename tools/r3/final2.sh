set -e
mkdir -p gpurun_out/r3/final
O=gpurun_out/r3/final
python3 -m pytest tests -x -q -m gpu > $O/t_full.log 2>&1 || { tail -40 $O/t_full.log; exit 1; }
tail -2 $O/t_full.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_config3_k20.json 2> $O/bench.err
python3 -c "
import json
d=json.load(open('$O/bench_config3_k20.json')); print(d['ms_per_step'], d['extra']['windows_ms_per_step'], d['single_proof_latency_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['valu_issue']['frac_of_ceiling'], d['h2d_inclusive']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['full_size'])"
python3 __graft_entry__.py --smoke 2>&1 | tail -1
