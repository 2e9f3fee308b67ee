set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
g++ -std=c++17 -pthread -O2 -I include examples/prove_stream.cpp -L tap-stark_amd/lib -ltapstark_hip -Wl,-rpath,$PWD/tap-stark_amd/lib -o /tmp/prove_stream
O=$R/gpurun_out/r3/kt_stream
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -o kt -- /tmp/prove_stream 20 16 4 device > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
cd $R
tail -1 $O/run.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3/kt_stream/**/kt_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
qs = collections.Counter(r.get("Queue_Id") for r in rows)
ss = collections.Counter(r.get("Stream_Id") for r in rows)
print("queues", qs.most_common(10)); print("streams", ss.most_common(10))
# overlap: total time where >= 2 kernels from different queues run
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur = 0; last = ev[0][0]; busy1 = 0; busy2 = 0
for t, d in ev:
    if cur >= 1: busy1 += t - last
    if cur >= 2: busy2 += t - last
    cur += d; last = t
print("time with >=1 kernel running: %.2f ms; with >=2: %.2f ms; span %.2f ms" % (busy1/1e6, busy2/1e6, (ev[-1][0]-ev[0][0])/1e6))
PY
