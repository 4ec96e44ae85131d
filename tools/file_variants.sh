#!/bin/bash
# Builds variants of ONE csrc file (compile flags) into libalq_<name>.so next to the package: the other objects come from the
# product build.   tools/file_variants.sh <file stem> name1 "flags1" name2 "flags2" ...
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$ROOT/nn-active-learning_amd/csrc"
STEM="$1"; shift
X=""; case "$STEM" in c3d|d3d|f3d|igemm4) X="-fno-slp-vectorize";; esac
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$CS -Wall -Wno-unused-function -Werror=extra-tokens $X"
while [ $# -ge 2 ]; do
  N="$1"; F="$2"; shift 2
  ( mkdir -p "$CS/build_v_$N"; hipcc $FLAGS $F -c "$CS/$STEM.hip" -o "$CS/build_v_$N/$STEM.o"
    OBJS=$(ls "$CS"/build/*.o | grep -v "/$STEM.o" | grep -v "hip-amdgcn")
    hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/nn-active-learning_amd/libalq_$N.so" $OBJS "$CS/build_v_$N/$STEM.o" -ldl
    echo "built libalq_$N.so ($STEM: $F)" ) &
done
wait
