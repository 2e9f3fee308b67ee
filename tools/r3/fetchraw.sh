set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3/fetchraw
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in config3 config4; do
  timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $O/$cfg -o pmc -- python3 $R/tools/prof_prove.py 2 $cfg > $O/$cfg.log 2>&1 || { echo "pmc $cfg failed"; tail -5 $O/$cfg.log; exit 1; }
  F=$(find $O/$cfg -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_fetch_raw.py $F 2 $cfg > $O/fetch_raw_$cfg.json
  python3 -c "
import json; d=json.load(open('$O/fetch_raw_$cfg.json')); print('$cfg', json.dumps(d['calibration'], indent=0))"
done
find $O -name "*.csv" -size +20M -delete
