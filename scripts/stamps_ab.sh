#!/bin/bash
# phase stamps of the banded kernel for several builds of the library (timing experiments)
for lib in "$@"; do
  echo "== $lib"
  BTF_LIB_PATH=$lib timeout -k 10 200 python scripts/stamps.py 2>&1 | grep -E "factor|backward|total"
done
