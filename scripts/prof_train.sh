#!/bin/bash
# rocprofv3 kernel statistics of the train-step bench -> gpurun_out/prof_r2/train (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r2/train
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2/train -- python3 $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline > $R/gpurun_out/prof_r2/train_bench.json 2> $R/gpurun_out/prof_r2/train.err
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_r2/train/*/*kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
steps = 26 + 1  # warm-up + timed + the instrumented one
for r in rows[:40]:
    print(f"{r['Name'][:100]:100s} calls/step={int(r['Calls'])/steps:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
print('total kernel ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps)
PY
