#!/bin/bash
# Profile job (run through gpurun):  scripts/gpu_profile_job.sh <tag> <config> <variant> [steps]
# rocprofv3 kernel stats and the two PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes on gfx950: TCC
# slots) of `bench.py --config <config> --variant <variant>`; then scripts/summarize_profiles.py <tag> <config> <variant>.
R=${GRAFT_REPO_ROOT:-/root/repo}
# EXTRA="--as-rank 0/8" SUFFIX=_rank0of8: further bench.py arguments and a suffix of the file stem
TAG=${1:-r03}; CFG=${2:-c3}; VAR=${3:-complete}; STEPS=${4:-100}
NAME=${TAG}_${CFG}_${VAR}${SUFFIX:-}
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
ARGS="--config $CFG --variant $VAR --no-cpu ${EXTRA:-}"
PARGS="$ARGS --no-c4"      # profiled passes: this variant's kernels only (the default run's short C4 leg stays out)
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$NAME -o $NAME --output-format csv -- python3 $R/bench.py $PARGS --steps $STEPS --warmup 10 > $R/gpurun_out/prof_$NAME.log 2>&1 || exit 1
echo "stats pass done"
# the same with --lean: nothing but W+V steps behind the burn-in, so that AverageNs of the accumulation kernels is over the
# dispatches bench.py's HIP events time (bench.py reads it back as roofline.rocprof_avg_us)
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${NAME}_lean -o ${NAME}_lean --output-format csv -- python3 $R/bench.py $PARGS --lean --steps $STEPS --warmup 10 > $R/gpurun_out/prof_${NAME}_lean.log 2>&1 || exit 1
echo "lean stats pass done"
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_$NAME -o $NAME --output-format csv -- python3 $R/bench.py $PARGS --steps 20 --warmup 2 --burn 2 > $R/gpurun_out/pmc_fetch_$NAME.log 2>&1 || exit 1
echo "fetch pass done"
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write_$NAME -o $NAME --output-format csv -- python3 $R/bench.py $PARGS --steps 20 --warmup 2 --burn 2 > $R/gpurun_out/pmc_write_$NAME.log 2>&1 || exit 1
echo "write pass done"
# occupancy / stall counters of the same command (their own pass)
timeout -k 10 900 rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_sq_$NAME -o $NAME --output-format csv -- python3 $R/bench.py $PARGS --steps 20 --warmup 2 --burn 2 > $R/gpurun_out/pmc_sq_$NAME.log 2>&1 || echo "sq pass failed (counters unavailable?)"
echo "sq pass done"
# the bench line kept beside the profile comes from a clean run of the same command: under rocprofv3 the HIP-event
# timings of bench.py carry the tool's per-dispatch overhead (its own AverageNs column does not)
cd $R && timeout -k 10 900 python3 $R/bench.py $ARGS --steps $STEPS --warmup 10 > $R/gpurun_out/bench_$NAME.log 2>/dev/null || exit 1
tail -1 $R/gpurun_out/bench_$NAME.log
