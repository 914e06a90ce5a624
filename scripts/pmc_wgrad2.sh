#!/bin/bash
# Where the waves of wgrad_grouped_kernel spend their cycles (scripts/bench_wgrad_grouped.py under rocprofv3 --pmc, counters in their own passes).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pw
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pw -- python3 $R/scripts/bench_wgrad_grouped.py > /dev/null 2>/tmp/pw.err
  python3 - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pw/*/*counter_collection.csv')
if not f:
    print("no counters for: $set"); print(open('/tmp/pw.err').read()[-400:])
else:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'wgrad_grouped_kernel<2, 4, 4>' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(f"{k:28s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
PY
done
