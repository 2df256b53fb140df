#!/bin/bash
# rows-per-workgroup sweep of one rank's share of C5 (bench.py --as-rank R/P, collectives in line): scripts/ab_rank.sh "W V" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for r in "$@"; do
  echo -n "rpb $r: "
  timeout -k 10 300 python bench.py --as-rank ${AS_RANK:-0/8} --steps 150 --no-cpu ${EXTRA:-} --rpb $r 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['projected']['per_rank_step_us'], d['kernels_us'], d['projected']['collective_us'])"
done
