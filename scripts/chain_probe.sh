#!/bin/bash
# builds and runs scripts/chain_probe.hip on the GPU box (plus variants given as extra -D flags, one run per argument)
mkdir -p gpurun_out
for v in "" "$@"; do
  echo "== variant: '$v'"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 $v -I functionalmf_amd/csrc -I include scripts/chain_probe.hip -o /tmp/chain_probe 2>/dev/null || exit 1
  timeout -k 10 120 /tmp/chain_probe || exit 1
done 2>&1 | tee gpurun_out/chain_probe.txt
