#!/bin/bash
# builds and runs scripts/pgx_probe.hip on the GPU box (plus variants given as extra -D flags, one run per argument)
set -e
mkdir -p gpurun_out
for v in "" "$@"; do
  echo "== variant: '$v'"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -I functionalmf_amd/csrc scripts/pgx_probe.hip -o /tmp/pgx_probe
  timeout -k 10 120 /tmp/pgx_probe
done 2>&1 | tee gpurun_out/pgx_probe.txt
