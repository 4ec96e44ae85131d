#!/bin/bash
# Builds variants of csrc/c3d.hip (compile flags) into libalq_<name>.so next to the package: the other objects come from the
# product build.   tools/c3b7_variants.sh name1 "flags1" name2 "flags2" ...
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$ROOT/nn-active-learning_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wall -Wno-unused-function -Werror=extra-tokens -fno-slp-vectorize"
while [ $# -ge 2 ]; do
  N="$1"; F="$2"; shift 2
  ( mkdir -p "$CS/build_v_$N"; hipcc $FLAGS $F -c "$CS/c3d.hip" -o "$CS/build_v_$N/c3d.o"
    OBJS=$(ls "$CS"/build/*.o | grep -v "/c3d.o" | grep -v "hip-amdgcn")
    hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/nn-active-learning_amd/libalq_$N.so" $OBJS "$CS/build_v_$N/c3d.o" -ldl
    echo "built libalq_$N.so ($F)" ) &
done
wait
