R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
python3 bench.py --workload asr_decode --steps 20 --warmup 3 > $O/bench_decode.json 2> $O/bench_decode.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/decode
rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode -- python3 $R/bench.py --workload asr_decode --steps 4 --warmup 2 --no-cpu-baseline > $O/decode_bench.json 2> $O/decode.err
ls $O/decode/*/ | head -3
