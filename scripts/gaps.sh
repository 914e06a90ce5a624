#!/bin/bash
# Kernel trace of the graph-replayed train step -> per-step busy / idle time of the device (union of kernel intervals over all streams)
# and the gap statistics between consecutive kernels.  Run through gpurun from the repo root.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/prof_r2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r2/gaps
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r2/gaps -- python3 $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline > $R/gpurun_out/prof_r2/gaps_bench.json 2> $R/gpurun_out/prof_r2/gaps.err
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_r2/gaps/*/*kernel_trace.csv'))[-1]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', ''))) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region: the last 20 adam_step launches delimit steps
adam = [i for i, r in enumerate(rows) if 'adam_step_kernel' in r[2]]
lo, hi = adam[-11], adam[-1]          # ten whole steps (adam end to adam end)
seg = rows[lo + 1: hi + 1]
t0, t1 = rows[lo][1], rows[hi][1]
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n, q in seg:
    if cur_e is None:
        cur_s, cur_e = s, e
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
steps = 10
print(f"wall {(t1 - t0) / 1e6 / steps:.3f} ms/step, device busy (union) {busy / 1e6 / steps:.3f} ms/step, idle {((t1 - t0) - busy) / 1e6 / steps:.3f} ms/step, "
      f"{len(seg) / steps:.0f} kernels/step, {len(gaps) / steps:.0f} gaps/step, mean gap {sum(g for g, _ in gaps) / max(1, len(gaps)) / 1e3:.2f} us")
import collections
h = collections.Counter()
for g, _ in gaps:
    h[min(g // 500, 20)] += 1
for k in sorted(h):
    print(f"  gap {k * 0.5:4.1f}-{k * 0.5 + 0.5:4.1f} us: {h[k] / steps:6.1f} per step")
big = sorted(gaps, reverse=True)[:15]
for g, n in big:
    print(f"  {g / 1e3:7.1f} us before {n[:90]}")
PY
