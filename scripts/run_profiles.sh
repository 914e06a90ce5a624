# produces the rocprofv3 summaries copied into profiles/ (run through gpurun from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/frontend -- python bench.py --workload frontend --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof/frontend_bench.json 2> gpurun_out/prof/frontend.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/train -- python bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/prof/train_bench.json 2> gpurun_out/prof/train.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/frontend_fetch -- python bench.py --workload frontend --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/prof/frontend_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/frontend_write -- python bench.py --workload frontend --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/prof/frontend_write.err
ls -R gpurun_out/prof | head -40
