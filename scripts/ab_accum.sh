cd $GRAFT_REPO_ROOT
run() { BTF_LIB_PATH=$GRAFT_REPO_ROOT/functionalmf_amd/libbtf_$1.so timeout -k 10 400 python bench.py --no-cpu ${@:2} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['kernels_us'])"; }
for v in w16_u4 w16_u2 w16_u6; do echo "== $v"; run $v --steps 150 --warmup 15; done
for r in "128 64" "128 128" "512 128" "512 256" "256 64" "1024 128"; do echo "== w16_u4 rpb $r"; run w16_u4 --steps 150 --warmup 15 --rpb $r; done
for v in w16_u4 w16_u2; do echo "== c5 $v"; run $v --config c5 --steps 15 --warmup 3 --burn 1; done
