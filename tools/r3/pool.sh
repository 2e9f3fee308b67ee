g++ -std=c++17 -pthread -O2 -I include examples/prove_stream.cpp -L tap-stark_amd/lib -ltapstark_hip -Wl,-rpath,$PWD/tap-stark_amd/lib -o /tmp/prove_stream
TS_POOL_DEBUG=1 /tmp/prove_stream 20 12 2 device 2>&1 | grep -v amdgpu.ids | awk '{print NR": "$0}' | tail -60
