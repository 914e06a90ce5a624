#!/bin/bash
# Kernel trace of beam-search decoding -> for one position in the middle of a replayed batch: wall time between two beam_advance launches, kernels by
# stream and name, idle time of the main stream.  Run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/decode
rocprofv3 --kernel-trace --output-format csv -d $O/decode -- python3 $R/bench.py --workload asr_decode --steps 3 --warmup 2 --no-cpu-baseline > $O/decode_bench.json 2> $O/decode.err
cd $R
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/timeline/decode/*/*kernel_trace.csv'))[-1]
rd = list(csv.DictReader(open(f)))
qk = 'Stream_Id' if 'Stream_Id' in rd[0] else 'Queue_Id'
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('(anonymous namespace)::', ''), r[qk]) for r in rd)
adv = [i for i, r in enumerate(rows) if 'advance_kernel' in r[2]]
print(len(adv), "positions in the trace")
# positions 40..60 of the last replayed batch but one (the very last batch is the instrumented eager one)
per = 82
base = len(adv) - 2 * per
lo, hi = adv[base + 40], adv[base + 60]
seg = rows[lo + 1: hi + 1]
wall = (rows[hi][1] - rows[lo][1]) / 20
print(f"wall per position {wall / 1e3:.1f} us over 20 positions, {len(seg) / 20:.1f} kernels per position, by {qk}")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in seg:
    by[(q, n[:100])][0] += 1
    by[(q, n[:100])][1] += e - s
for q in sorted({k[0] for k in by}):
    tot = sum(v[1] for k, v in by.items() if k[0] == q)
    # idle time of the stream: gaps between consecutive kernels of this stream
    ks = [(s, e) for s, e, n, qq in seg if qq == q]
    idle = sum(max(0, ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1))
    big = sorted(((ks[i + 1][0] - ks[i][1]) for i in range(len(ks) - 1)), reverse=True)[:25]
    print(f"-- {qk} {q}: {len(ks) / 20:.1f} kernels, busy {tot / 20e3:.1f} us, idle between its kernels {idle / 20e3:.1f} us per position; largest gaps (us): {[round(b / 1e3, 1) for b in big]}")
    for (qq, n), v in sorted(by.items(), key=lambda kv: -kv[1][1]):
        if qq == q and v[1] / 20e3 >= 4:
            print(f"   {v[1] / 20e3:8.1f} us {v[0] / 20:5.1f} x {v[1] / v[0] / 1e3:7.1f}  {n}")
with open('gpurun_out/timeline/decode_sequence.txt', 'w') as g:
    lo2, hi2 = adv[base + 50], adv[base + 51]
    last = {}
    for s, e, n, q in rows[lo2 + 1: hi2 + 1]:
        gap = (s - last[q]) / 1e3 if q in last else 0.0
        last[q] = e
        g.write(f"{(s - rows[lo2][1]) / 1e3:9.1f} q{q} gap {gap:6.1f} dur {(e - s) / 1e3:7.1f}  {n[:120]}\n")
PY
