set -e
mkdir -p gpurun_out/r3/final
O=gpurun_out/r3/final
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_config3_k20.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_config3_k20_b.json 2>> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_config3_k20_c.json 2>> $O/bench.err
python3 bench.py --workload config2 --streams 8 --steps 48 --warmup 8 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
python3 bench.py --workload config4 --streams 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > $O/bench_config4.json 2>> $O/bench.err
python3 bench.py --workload config4 --streams 2 --steps 8 --warmup 2 --windows 1 --no-cpu-baseline --headline-only > $O/bench_config4_2lanes.json 2>> $O/bench.err
python3 bench.py --workload config5 --streams 2 --steps 8 --warmup 2 --windows 2 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
python3 bench.py --workload config4 --mode sharded --steps 4 --warmup 1 --windows 1 --no-cpu-baseline --headline-only > $O/bench_config4_sharded_world1.json 2>> $O/bench.err
TS_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 8 --warmup 2 --headline-only > $O/bench_gpus2_shared_gpu_rehearsal.json 2>> $O/bench.err
python3 tools/shard_stages.py 22 8 > $O/config4_shard_stages.json 2>> $O/bench.err
TS_SOAK_PROOFS=1500 python3 -m pytest tests/test_gpu_soak.py -q -m gpu > $O/soak_long.log 2>&1 || { tail -5 $O/soak_long.log; exit 1; }
tail -1 $O/soak_long.log
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3/final/bench_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["ms_per_step"], 4), d.get("extra", {}).get("windows_ms_per_step"), d.get("single_proof_latency_ms"))
PY
