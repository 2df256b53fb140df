#!/bin/bash
# Round profile job (run through gpurun): tests, bench, rocprofv3 kernel stats and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes on gfx950: TCC slots).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.log 2>&1; tail -1 gpurun_out/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python $R/bench.py --steps 100 --warmup 10 --no-cpu > $R/gpurun_out/prof_$TAG.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_$TAG -o $TAG --output-format csv -- python $R/bench.py --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write_$TAG -o $TAG --output-format csv -- python $R/bench.py --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/pmc_write_$TAG.log 2>&1
ls $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_fetch_$TAG $R/gpurun_out/pmc_write_$TAG
