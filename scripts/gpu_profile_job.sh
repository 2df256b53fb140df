set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/bench2.log 2>&1; tail -1 gpurun_out/bench2.log
for r in "64 64" "128 64" "256 128" "512 128" "1024 256" "2048 512"; do echo "rpb $r"; timeout -k 10 200 python bench.py --no-cpu --steps 100 --warmup 10 --rpb $r 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernels_us'])"; done > gpurun_out/rpb_sweep.log 2>&1
cat gpurun_out/rpb_sweep.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2 -o r1 --output-format csv -- python $R/bench.py --steps 50 --warmup 5 --no-cpu > $R/gpurun_out/prof2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o r1 --output-format csv -- python $R/bench.py --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write -o r1 --output-format csv -- python $R/bench.py --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/pmc_write.log 2>&1
ls $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
