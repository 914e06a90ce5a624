#!/bin/bash
# Kernel trace of the graph-replayed step of one bench workload -> per stream, the kernels of ONE replayed step by name (launches, total and
# mean duration) and the whole ordered sequence (gpurun_out/timeline/<workload>_sequence.txt).  Run through gpurun from the repo root:
#   scripts/timeline.sh [asr_train_step|tts_train_step]
R=${GRAFT_REPO_ROOT:-$PWD}
W=${1:-asr_train_step}
O=$R/gpurun_out/timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/$W
rocprofv3 --kernel-trace --output-format csv -d $O/$W -- python3 $R/bench.py --workload $W --steps 12 --warmup 4 --no-cpu-baseline > $O/${W}_bench.json 2> $O/${W}.err
cd $R
python3 - $W <<'PY'
import csv, glob, sys, collections
W = sys.argv[1]
f = sorted(glob.glob(f'gpurun_out/timeline/{W}/*/*kernel_trace.csv'))[-1]
rd = list(csv.DictReader(open(f)))
qk = 'Stream_Id' if 'Stream_Id' in rd[0] else 'Queue_Id'
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', ''), r[qk]) for r in rd)
adam = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r[2]]
lo, hi = adam[-4], adam[-3]            # one whole replayed step (the last launch of the trace is the instrumented eager step)
seg = rows[lo + 1: hi + 1]
print(f"{W}: one step = {(rows[hi][1] - rows[lo][1]) / 1e3:.1f} us, {len(seg)} kernels, by {qk}")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in seg:
    by[(q, n[:110])][0] += 1
    by[(q, n[:110])][1] += e - s
for q in sorted({k[0] for k in by}):
    tot = sum(v[1] for k, v in by.items() if k[0] == q)
    print(f"-- {qk} {q}: {sum(v[0] for k, v in by.items() if k[0] == q)} kernels, {tot / 1e3:.1f} us")
    for (qq, n), v in sorted(by.items(), key=lambda kv: -kv[1][1]):
        if qq == q and v[1] / 1e3 >= 15:
            print(f"   {v[1] / 1e3:8.1f} us {v[0]:4d} x {v[1] / v[0] / 1e3:7.1f}  {n}")
with open(f'gpurun_out/timeline/{W}_sequence.txt', 'w') as g:
    last = {}
    for s, e, n, q in seg:
        gap = (s - last[q]) / 1e3 if q in last else 0.0
        last[q] = e
        g.write(f"{(s - rows[lo][1]) / 1e3:9.1f} q{q} gap {gap:6.1f} dur {(e - s) / 1e3:7.1f}  {n[:140]}\n")
PY
