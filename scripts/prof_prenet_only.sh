# per-kernel stats of the prenet alone at both widths (run through gpurun from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out/pn
for d in 64,64 512,512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pn/$d -- python scripts/bench_prenet.py $d > gpurun_out/pn/$d.log 2> gpurun_out/pn/$d.err
done
python - <<'PY'
import csv, glob, os
for d in ("64,64", "512,512"):
    fs = sorted(glob.glob(f"gpurun_out/pn/{d}/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(fs[-1])))
    print(d, open(f"gpurun_out/pn/{d}.log").read().strip())
    for r in rows[:14]:
        print(f"  {r['Name'][:84]:84s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
