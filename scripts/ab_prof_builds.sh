R=$PWD
cd /tmp && export TMPDIR=/tmp
for d in new old; do
  if [ $d = new ]; then T=$R; else T=$R/_ab_old; fi
  rm -rf /tmp/p_$d
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$d -- python3 $T/bench.py --steps 30 --warmup 6 --no-cpu-baseline > /tmp/p_$d.json 2>/dev/null
done
python3 - <<'PY'
import csv, glob
def load(d):
    f = glob.glob(f'/tmp/p_{d}/*/*kernel_stats.csv')[0]
    return {r['Name'].replace('(anonymous namespace)::','')[:80]: (int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3) for r in csv.DictReader(open(f))}
a, b = load('new'), load('old')
tot_a = sum(v[2] for v in a.values()); tot_b = sum(v[2] for v in b.values())
print(f"total kernel time per 37 steps: new {tot_a/37:.1f} us/step, old {tot_b/37:.1f} us/step")
rows = []
for k in a:
    if k in b:
        rows.append((a[k][2] - b[k][2], k, a[k], b[k]))
for d, k, x, y in sorted(rows, key=lambda r: -abs(r[0]))[:16]:
    print(f"{d/37:8.1f} us/step  new {x[1]:8.2f} old {y[1]:8.2f} us x {x[0]/37:.0f}  {k}")
PY
