set -e
mkdir -p gpurun_out/r3
g++ -std=c++17 -pthread -O2 -I include examples/prove_stream.cpp -L tap-stark_amd/lib -ltapstark_hip -Wl,-rpath,$PWD/tap-stark_amd/lib -o /tmp/prove_stream
echo "--- system HIP runtime"; /tmp/prove_stream 20 40 4 device
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
echo "--- torch's HIP runtime preloaded"; LD_LIBRARY_PATH=$TL LD_PRELOAD="$TL/libamdhip64.so" /tmp/prove_stream 20 40 4 device || true
echo "--- GPU_MAX_HW_QUEUES=8"; GPU_MAX_HW_QUEUES=8 /tmp/prove_stream 20 40 4 device
echo "--- bench.py with the system runtime (TS_PRELOAD_TORCH=0)"
TS_PRELOAD_TORCH=0 python3 bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['extra']['windows_ms_per_step'])" || true
echo "--- bench.py default"
python3 bench.py --steps 20 --warmup 5 --headline-only 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['extra']['windows_ms_per_step'])"
