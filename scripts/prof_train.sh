# per-kernel stats of the default train-step bench (run through gpurun from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train8 -- python bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/prof_train8.json 2> gpurun_out/prof_train8.err
