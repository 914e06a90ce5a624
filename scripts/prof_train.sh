# per-kernel stats of the default train-step bench (run through gpurun from the repo root); extra env is passed through
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-prof_train}
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$OUT -- python bench.py --steps 12 --warmup 6 --no-cpu-baseline > gpurun_out/$OUT.json 2> gpurun_out/$OUT.err
