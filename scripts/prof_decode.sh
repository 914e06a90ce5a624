#!/bin/bash
# rocprofv3 kernel statistics of the decode bench -> gpurun_out/prof_r2/decode (run through gpurun from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/prof_r2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r2/decode
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2/decode -- python3 $R/bench.py --workload asr_decode --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_r2/decode_bench.json 2> $R/gpurun_out/prof_r2/decode.err
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_r2/decode/*/*kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
steps = 3 + 1 + 1  # warm-up + timed + the instrumented one
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} calls/batch={int(r['Calls'])/steps:8.1f} ms/batch={float(r['TotalDurationNs'])/1e6/steps:8.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
print('total kernel ms/batch', sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps)
PY
