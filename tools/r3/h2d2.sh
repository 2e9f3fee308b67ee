g++ -std=c++17 -pthread -O2 -I include examples/prove_stream.cpp -L tap-stark_amd/lib -ltapstark_hip -Wl,-rpath,$PWD/tap-stark_amd/lib -o /tmp/prove_stream
/tmp/prove_stream 20 24 4 pinned 2>&1 | grep -v amdgpu.ids
/tmp/prove_stream 20 80 4 pinned 2>&1 | grep -v amdgpu.ids
/tmp/prove_stream 20 40 2 pinned 2>&1 | grep -v amdgpu.ids
/tmp/prove_stream 20 40 8 pinned 2>&1 | grep -v amdgpu.ids
