cd /tmp && export TMPDIR=/tmp
for cfg in "8 1024" "4 2048" "4 4096" "2 4096" "16 512"; do
  set -- $cfg
  rm -rf /tmp/ln; SC_LN_BWD_ROWS=$1 SC_LN_BWD_CAP=$2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ln -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 --no-cpu-baseline > /tmp/ln.json 2>/dev/null
  python3 -c "
import csv,glob,json
for r in csv.DictReader(open(glob.glob('/tmp/ln/*/*kernel_stats.csv')[0])):
    if any(k in r['Name'] for k in ('layernorm_bwd_vec','reduce_parts_batched')): print('$cfg', r['Name'][28:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
"
done
