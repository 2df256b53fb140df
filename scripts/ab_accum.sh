# A/B of several builds of the library in one gpurun call:  bash scripts/ab_accum.sh <lib> <lib> ...  (names after libbtf_)
cd $GRAFT_REPO_ROOT
run() { BTF_LIB_PATH=$GRAFT_REPO_ROOT/functionalmf_amd/libbtf_$1.so timeout -k 10 400 python bench.py --no-cpu ${@:2} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['kernels_us'])"; }
for rep in ${AB_REPS:-1 2}; do
for v in "$@"; do
  echo "== $v complete";  run $v --steps 150 --warmup 15 --variant complete
  echo "== $v missing5";  run $v --steps 150 --warmup 15 --variant missing5
  echo "== $v heldout";  run $v --steps 150 --warmup 15 --variant heldout
  echo "== $v binomial";  run $v --steps 60 --warmup 10 --variant binomial
done; done
