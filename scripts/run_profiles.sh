#!/bin/bash
# Produces everything scripts/collect_profiles.py copies into profiles/ for one round (run through gpurun from the repo root, ONE call):
# the bench lines (train step with both prenets, frontend, FastSpeech2, beam-search decoding), the rocprofv3 kernel statistics of the four workloads, and the
# FETCH_SIZE / WRITE_SIZE passes (counters in their own runs, --kernel-trace only, as the MI355X guide prescribes).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
echo "== bench lines"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --no-cpu-baseline --prenet-dims 512,512 > $O/bench_p512.json 2> $O/bench_p512.err
python3 bench.py --workload frontend > $O/bench_frontend.json 2> $O/bench_frontend.err
python3 bench.py --workload tts_train_step > $O/bench_tts.json 2> $O/bench_tts.err
python3 bench.py --workload asr_decode --steps 20 --warmup 3 > $O/bench_decode.json 2> $O/bench_decode.err
cd /tmp && export TMPDIR=/tmp
echo "== kernel statistics"
rm -rf $O/train $O/frontend $O/tts $O/decode $O/train_mfma $O/train_fetch $O/train_write $O/frontend_fetch $O/frontend_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py --steps 20 --warmup 6 --no-cpu-baseline > $O/train_bench.json 2> $O/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/frontend -- python3 $R/bench.py --workload frontend --steps 30 --warmup 5 --no-cpu-baseline > $O/frontend_bench.json 2> $O/frontend.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tts -- python3 $R/bench.py --workload tts_train_step --steps 20 --warmup 6 --no-cpu-baseline > $O/tts_bench.json 2> $O/tts.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/decode -- python3 $R/bench.py --workload asr_decode --steps 4 --warmup 2 --no-cpu-baseline > $O/decode_bench.json 2> $O/decode.err
echo "== counters"
export BENCH_NO_GRAPH=1 BENCH_NO_PREFETCH=1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/train_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/train_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/train_write -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/train_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/train_mfma -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/train_mfma.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/frontend_fetch -- python3 $R/bench.py --workload frontend --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/frontend_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/frontend_write -- python3 $R/bench.py --workload frontend --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> $O/frontend_write.err
ls $O
