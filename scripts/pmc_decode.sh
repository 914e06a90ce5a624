#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the beam-search decoding bench in two separate --pmc passes (--kernel-trace only, as the MI355X guide prescribes)
# -> gpurun_out/prof/decode_fetch, decode_write; scripts/collect_pmc_decode.py turns them into profiles/<round>_decode_pmc_traffic.json.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/decode_fetch $O/decode_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/decode_fetch -- python3 $R/bench.py --workload asr_decode --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/decode_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/decode_write -- python3 $R/bench.py --workload asr_decode --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/decode_write.err
ls $O/decode_fetch/*/ | head -3
