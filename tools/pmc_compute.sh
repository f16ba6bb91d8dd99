#!/bin/bash
# dev tool (GPU box): compute-side PMC passes over a few cfg3-P2 steps (one rocprofv3 run per counter group, --pmc with --kernel-trace only):
# per kernel VALU / LDS / memory-unit utilisation -> gpurun_out/pmcc_<group>.csv summaries
#   tools/pmc_compute.sh [p2|p1]
variant=${1:-p2}
R=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $R/gpurun_out/pmcc_avail.txt 2>&1
i=0
for grp in "VALUBusy SALUBusy" "MemUnitStalled OccupancyPercent" "WriteUnitStalled LDSBankConflict" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum MeanOccupancyPerCU"; do
  i=$((i+1)); O=$R/gpurun_out/pmcc_$i; rm -rf $O
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O -o c -- python3 $R/bench.py --variant $variant --steps 3 --warmup 1 --steady-steps 0 --no-dp-probe --no-cpu-baseline --no-roofline --no-variants > $O.log 2>&1 || echo "group $i ($grp) failed" 
  python3 - "$O" "$grp" <<'PY' >> $R/gpurun_out/pmcc_summary_$variant.txt
import csv, glob, sys, collections
d, grp = sys.argv[1], sys.argv[2]
fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
if not fs: print('no data for', grp); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(fs[0])):
    if 'aefft' not in r['Kernel_Name']: continue
    k = r['Kernel_Name'].replace('void ', '').replace('aefft::', ''); k = k[:k.index('(')] if '(' in k else k
    a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
print('== ' + grp)
for k in sorted(acc):
    print(f"  {k[:44]:44s} " + "  ".join(f"{c}={v[1]/max(v[0],1):.4g} (n={v[0]})" for c, v in sorted(acc[k].items())))
PY
done
