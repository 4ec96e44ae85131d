#!/bin/bash
# builds and times variants of the plane-sweep kernels on the GPU box:  tools/probe/run_variants.sh "<flags 1>" "<flags 2>" ...
cd "$(dirname "${BASH_SOURCE[0]}")/../.."
i=0
for F in "$@"; do
  i=$((i+1))
  if hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -w -Iinclude -Inn-active-learning_amd/csrc $F tools/probe/c3d_bench.hip -o /tmp/c3b_$i 2>/tmp/c3b_$i.err; then
    echo "== [$F]"; /tmp/c3b_$i ${C3_N:-2000}
  else
    echo "== [$F] BUILD FAILED"; tail -5 /tmp/c3b_$i.err
  fi
done
