#!/bin/bash
# PMC passes of the grouped weight-gradient kernel alone (scripts/bench_wgrad_grouped.py): L2 hit rate and bytes fetched beyond L2.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for which in ffn1 sq; do
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc_wg_hit_$which --output-format csv -- python3 $R/scripts/bench_wgrad_grouped.py $which > $R/gpurun_out/pmc_wg_hit_$which.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_wg_fetch_$which --output-format csv -- python3 $R/scripts/bench_wgrad_grouped.py $which > $R/gpurun_out/pmc_wg_fetch_$which.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_wg_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            cnt[(k, r['Counter_Name'])] += 1
        for k, v in acc.items():
            if 'wgrad' in k:
                print(d, k, {c: (x / cnt[(k, c)]) for c, x in v.items()})
PY
