#!/bin/bash
# Builds libalq.so (gfx950 only) next to the Python package.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="$(dirname "$HERE")"
ROOT="$(dirname "$PKG")"
OUT="$PKG/${ALQ_OUT:-libalq.so}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$HERE -Wall -Wno-unused-function"
BUILD="$HERE/build${ALQ_BUILD_TAG:-}"
mkdir -p "$BUILD"
pids=()
for f in igemm igemm2 igemm3 igemm4 fcgemm direct kernels topk model comm train sim; do
  ( hipcc $FLAGS -c "$HERE/$f.hip" -o "$BUILD/$f.o" ${ALQ_EXTRA_FLAGS:-} ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT" "$BUILD"/{igemm,igemm2,igemm3,igemm4,fcgemm,direct,kernels,topk,model,comm,train,sim}.o -ldl
echo "built $OUT"
