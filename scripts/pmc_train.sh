# HBM traffic of the train step's GEMM kernels: separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a few EAGER steps
# (run through gpurun from the repo root; counters in their own runs with --kernel-trace only, as the MI355X guide prescribes)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out/prof
export BENCH_NO_GRAPH=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/train_fetch -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/prof/train_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/train_write -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/prof/train_write.err
ls gpurun_out/prof/train_fetch/*/ | head
