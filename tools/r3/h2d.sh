g++ -std=c++17 -pthread -O2 -I include examples/prove_stream.cpp -L tap-stark_amd/lib -ltapstark_hip -Wl,-rpath,$PWD/tap-stark_amd/lib -o /tmp/prove_stream
/tmp/prove_stream 20 40 4 pinned 2>&1 | grep -v amdgpu.ids
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['h2d_inclusive'])"
/tmp/prove_stream 20 40 4 pinned 2>&1 | grep -v amdgpu.ids
