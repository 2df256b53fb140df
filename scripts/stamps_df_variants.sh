#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for tag in "$@"; do
  for df in 1 1; do
    echo "=== lib $tag dataflow=$df"
    BTF_VF_DATAFLOW=$df BTF_LIB_PATH=$PWD/functionalmf_amd/libbtf_$tag.so BTF_ACC_STAMPS_OUT=gpurun_out/st_${tag}_$df.npz timeout -k 10 300 python bench.py --steps 100 --no-cpu --no-c4 --lean 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels_us'])"
    python scripts/acc_stamps.py gpurun_out/st_${tag}_$df.npz | grep -A12 "^v accumulation" | grep -v "start  \|streaming"
  done
done
