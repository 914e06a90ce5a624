# bench lines + per-kernel stats of the train step with both prenet widths (run through gpurun from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cd $R
mkdir -p gpurun_out/pp
python bench.py --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/pp/b64.json 2> gpurun_out/pp/b64.err
python bench.py --steps 40 --warmup 6 --no-cpu-baseline --prenet-dims 512,512 > gpurun_out/pp/b512.json 2> gpurun_out/pp/b512.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp/t64 -- python bench.py --steps 12 --warmup 6 --no-cpu-baseline > gpurun_out/pp/t64.json 2> gpurun_out/pp/t64.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp/t512 -- python bench.py --steps 12 --warmup 6 --no-cpu-baseline --prenet-dims 512,512 > gpurun_out/pp/t512.json 2> gpurun_out/pp/t512.err
