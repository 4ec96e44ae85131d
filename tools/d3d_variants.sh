#!/bin/bash
# Builds libalq variants that differ in d3d.hip's compile flags (here, cross-compiled) and, on the GPU box, times the two d3d kernels in each.
#   build:  bash tools/d3d_variants.sh build "<flags v1>" "<flags v2>" ...     -> nn-active-learning_amd/libalq_v<i>.so
#   run:    bash tools/d3d_variants.sh run <count>
set -e
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"
if [ "$1" = build ]; then
  shift; i=0
  for fl in "$@"; do
    i=$((i+1))
    ALQ_BUILD_TAG=_v$i ALQ_OUT=libalq_v$i.so ALQ_D3_FLAGS="$fl" bash nn-active-learning_amd/csrc/build.sh | tail -1
  done
else
  N=$2; mkdir -p gpurun_out; export TMPDIR=/tmp
  for i in 0 $(seq 1 $N); do
    L=libalq_v$i.so; [ $i = 0 ] && L=libalq.so
    ( cd /tmp && ALQ_LIB=$L rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/d3v$i -o s --output-format csv -- python3 $ROOT/tools/gpu_d3d_time.py > /dev/null 2>&1 )
    echo "variant $i: $(grep -E '(d3d_fwd|d3d_bwd|f3d_fwd)' gpurun_out/d3v$i/s_kernel_stats.csv | awk -F, '{printf "%s %.1f us   ", substr($1,7,14), $4/1000}')"
  done
fi
