#!/bin/bash
# Instruction-mix counters of one workload's kernels:  scripts/pmc_kernel.sh <variant> "<counters>" [config]
R=${GRAFT_REPO_ROOT:-/root/repo}; VAR=${1:-binomial}; PMC=${2:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES"}; CFG=${3:-c3}
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc $PMC --kernel-trace -d $R/gpurun_out/pmck_$VAR -o pmck --output-format csv -- python3 $R/bench.py --config $CFG --variant $VAR --no-cpu --steps 10 --warmup 2 --burn 2 > $R/gpurun_out/pmck_$VAR.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmck_$VAR/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "pg" in k or "accum" in k or "twist" in k or "w_solve" in k:
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
