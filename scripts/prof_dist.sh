# kernel trace of the one-rank rehearsal of the multi-rank step (run through gpurun from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
export BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pd -- python bench.py --gpus 1 --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/pd.json 2> gpurun_out/pd.err
python - <<'PY'
import csv, glob, os, json
print(json.load(open("gpurun_out/pd.json"))["ms_per_step"])
fs = sorted(glob.glob("gpurun_out/pd/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(fs[-1])):
    if any(k in r["Name"].lower() for k in ("rccl", "nccl", "allreduce", "allgather", "broadcast", "memcpy", "copy")):
        print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us total/26 {float(r['TotalDurationNs'])/1e6/26:7.3f} ms")
PY
