#!/bin/bash
# Where the waves of the NT GEMM kernels spend their cycles, next to the vendor library's kernels on the same shapes
# (scripts/gemm_pmc_run.py under rocprofv3 --pmc, counters in their own passes).  Per kernel name and grid: averages over launches.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  rm -rf /tmp/pg
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pg -- python3 $R/scripts/gemm_pmc_run.py > /dev/null 2>/tmp/pg.err
  python3 - <<PY
import csv, glob, collections
f = glob.glob('/tmp/pg/*/*counter_collection.csv')
if not f:
    print("no counters for: $set"); print(open('/tmp/pg.err').read()[-600:])
else:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        name = r['Kernel_Name']
        if 'gemm' in name.lower() or 'Cijk' in name:
            key = (name[:70], r.get('Grid_Size', ''), r['Counter_Name'])
            agg[key].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(f"{k[0]:70s} grid {k[1]:>8s} {k[2]:32s} {sum(v)/len(v):16.0f}  ({len(v)})")
PY
done
