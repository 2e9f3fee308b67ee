set -e
g++ -std=c++17 -pthread -O2 -I include examples/prove_stream.cpp -L tap-stark_amd/lib -ltapstark_hip -Wl,-rpath,$PWD/tap-stark_amd/lib -o /tmp/prove_stream
echo "--- ctx per thread"; /tmp/prove_stream 20 40 4 device
echo "--- ctx in main"; PS_CTX_IN_MAIN=1 /tmp/prove_stream 20 40 4 device
echo "--- 2 lanes"; PS_CTX_IN_MAIN=1 /tmp/prove_stream 20 40 2 device
echo "--- 8 lanes"; PS_CTX_IN_MAIN=1 /tmp/prove_stream 20 40 8 device
