#!/bin/bash
# rocprofv3 kernel statistics of the FastSpeech2 train-step bench -> gpurun_out/prof_r2/tts
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/prof_r2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r2/tts
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2/tts -- python3 $R/bench.py --workload tts_train_step --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_r2/tts_bench.json 2> $R/gpurun_out/prof_r2/tts.err
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_r2/tts/*/*kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
steps = 26
for r in rows[:45]:
    print(f"{r['Name'][:100]:100s} calls/step={int(r['Calls'])/steps:7.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
print('total kernel ms/step', sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps, 'launches/step', sum(int(r['Calls']) for r in rows) / steps)
PY
