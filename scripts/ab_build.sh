#!/bin/bash
# A/B builds of the library:  scripts/ab_build.sh name "-DBTF_X=1 -DBTF_Y=2" [name2 "..."]  ->  functionalmf_amd/libbtf_<name>.so
# (run here, in the build container; the .so files travel to the GPU box with the snapshot; scripts/ab_run.sh runs them)
set -e
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  echo "== building libbtf_$1.so with '$2'"
  BTF_LIB_PATH=$PWD/functionalmf_amd/libbtf_$1.so BTF_BUILD_DEFS="$2" python -c "from functionalmf_amd import _native; _native.build(force=True)" 2>&1 | grep -v warning | tail -3
  shift 2
done
