# rows-per-workgroup sweep of the accumulation launches:  AB_LIBS=.. AB_VARIANTS=.. AB_RPBS="0:0 512:256" bash scripts/ab_rpb.sh
cd $GRAFT_REPO_ROOT
run() { BTF_LIB_PATH=$GRAFT_REPO_ROOT/functionalmf_amd/libbtf_$1.so timeout -k 10 400 python bench.py --no-cpu ${@:2} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['kernels_us'])"; }
for lib in ${AB_LIBS:-hip}; do
for rpb in ${AB_RPBS:-0:0 0:256 0:512 512:256 128:256}; do
  for v in ${AB_VARIANTS:-complete missing5 binomial}; do
  echo "== $lib rpb $rpb $v"; run $lib --steps 100 --warmup 10 --variant $v --rpb ${rpb%%:*} ${rpb##*:}
  done
done; done
