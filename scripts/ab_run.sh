#!/bin/bash
# runs `python bench.py --no-cpu <args>` once per A/B library (names after libbtf_, "hip" = the default build):
#   AB_LIBS="hip k10w16 k10u4" scripts/ab_run.sh --config c3k10 --steps 200
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for rep in ${AB_REPS:-1 2}; do
for v in ${AB_LIBS:-hip}; do
  echo -n "== $v: "
  BTF_LIB_PATH=$PWD/functionalmf_amd/libbtf_$v.so timeout -k 10 400 python bench.py --no-cpu "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['kernels_us'])"
done; done
