#!/bin/bash
# SQ counters of the banded sampler (one rocprofv3 --pmc pass per counter group).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmcb_$i -o b --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/pmcb_$i.log 2>&1
  f=$(find $R/gpurun_out/pmcb_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "v_banded" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-24s per launch %.3e  (%d launches)" % (k, s / max(n, 1), n))
PY
done
