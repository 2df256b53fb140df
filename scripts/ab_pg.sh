# A/B of library builds on the Binomial workload:  bash scripts/ab_pg.sh <lib> <lib> ...  (names after libbtf_)
cd $GRAFT_REPO_ROOT
run() { BTF_LIB_PATH=$GRAFT_REPO_ROOT/functionalmf_amd/libbtf_$1.so timeout -k 10 400 python bench.py --no-cpu --variant binomial --steps 200 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['kernels_us'])"; }
for rep in 1 2 3; do for v in "$@"; do echo "== $v"; run $v; done; done
