set -e
mkdir -p gpurun_out/r3
python3 -m pytest tests/test_gpu_parity.py -x -q -k "cpp" > gpurun_out/r3/t_stream.log 2>&1 || { tail -30 gpurun_out/r3/t_stream.log; exit 1; }
tail -2 gpurun_out/r3/t_stream.log
g++ -std=c++17 -pthread -O2 -I include examples/prove_stream.cpp -L tap-stark_amd/lib -ltapstark_hip -Wl,-rpath,$PWD/tap-stark_amd/lib -o /tmp/prove_stream
/tmp/prove_stream 20 40 4 device | tee gpurun_out/r3/stream_device.txt
/tmp/prove_stream 20 40 4 pinned | tee gpurun_out/r3/stream_pinned.txt
/tmp/prove_stream 20 40 1 device | tee gpurun_out/r3/stream_device1.txt
