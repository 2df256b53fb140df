#!/bin/bash
# timeline of the accumulation workgroups (diagnostic build: scripts/ab_build.sh stamps "-DBTF_ACC_STAMPS"):
#   scripts/acc_stamps.sh tag [bench.py arguments]     -> gpurun_out/stamps_<tag>.npz, summary on stdout
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
tag=$1; shift
mkdir -p gpurun_out
echo "== $tag: $*"
BTF_LIB_PATH=$PWD/functionalmf_amd/libbtf_stamps.so BTF_ACC_STAMPS_OUT=gpurun_out/stamps_$tag.npz timeout -k 10 400 python bench.py --steps 100 --no-cpu "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels_us'])"
python scripts/acc_stamps.py gpurun_out/stamps_$tag.npz
