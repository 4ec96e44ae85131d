#!/bin/bash
# Builds libalq.so (gfx950 only) next to the Python package.  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
PKG="$(dirname "$HERE")"
ROOT="$(dirname "$PKG")"
OUT="$PKG/${ALQ_OUT:-libalq.so}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -I$HERE -Wall -Wno-unused-function -Werror=extra-tokens"
BUILD="$HERE/build${ALQ_BUILD_TAG:-}"
mkdir -p "$BUILD"
HAZARD_FILES="c3d d3d f3d t3d t3d8b e3d"
pids=()
for f in igemm igemm2 igemm3 igemm4 c3d t3d t3d8b e3d d3d f3d fcgemm direct kernels topk model comm train sim ref64; do
  # igemm4: no SLP vectorisation (a performance choice) - it turns neighbouring scalar f32 multiplies / adds of the staging and epilogue code into
  # v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, which issue slower than the scalar pairs next to another wave's MFMAs on
  # the same SIMD (same-box A/B: 155.4 k -> 156.6 k patches/s); conversions still pack (v_cvt_pk_f16_f32 / _bf16_f32)
  X=""; if [ "$f" = igemm4 ]; then X="-fno-slp-vectorize --save-temps=obj ${ALQ_G4_FLAGS:-}"; fi
  # c3d: the same choice (its staging / epilogue arithmetic runs between the wave's own MFMAs)
  if [ "$f" = c3d ]; then X="-fno-slp-vectorize ${ALQ_C3_FLAGS:-}"; fi
  if [ "$f" = d3d ] || [ "$f" = f3d ]; then X="-fno-slp-vectorize ${ALQ_D3_FLAGS:-}"; fi      # (d3d: like c3d; same-box 523 -> 490 / 710 -> 675 us)
  # the sweep kernels leave their device assembly behind for the store-hazard gate below
  case " $HAZARD_FILES " in *" $f "*) X="$X --save-temps=obj";; esac
  ( hipcc $FLAGS $X -c "$HERE/$f.hip" -o "$BUILD/$f.o" ${ALQ_EXTRA_FLAGS:-} ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
# Performance gate (not a correctness one: round 2 showed packed and scalar builds give bit-identical results,
# profiles/r02_fcf_diag.txt): the two-slot engine should hold no SLP-packed fp32 multiplies / fmas - beside another wave's
# MFMAs on the same SIMD they issue slower than the scalar pair (same-box A/B 155.4 k -> 156.6 k patches/s).  Checked on
# the device assembly the compile leaves behind; v_pk_add_f32 is the explicit vector add of the accumulate path.
ASM="$BUILD/igemm4-hip-amdgcn-amd-amdhsa-gfx950.s"
if [ -f "$ASM" ]; then
  if grep -E -q "v_pk_(mul|fma)_f32" "$ASM"; then
    echo "build.sh: packed fp32 arithmetic in igemm4 device code (slower next to MFMAs; see the note above):" >&2
    grep -E -n "v_pk_(mul|fma)_f32" "$ASM" | head -5 >&2
    exit 1
  fi
  rm -f "$BUILD"/igemm4-hip-*.bc "$BUILD"/igemm4-hip-*.hipi "$BUILD"/igemm4-host-*.bc "$BUILD"/igemm4-host-*.hipi "$BUILD"/igemm4-host-*.s
else
  # another ROCm version may name the --save-temps files differently: the check is a tuning aid, not a build requirement
  echo "build.sh: warning: device assembly of igemm4 not found ($ASM); packed-fp32 check skipped" >&2
fi
# Correctness gate: the 16-byte-store data hazard (alq_internal.h, ALQ_STORE_HOLD).  The compiler inserts no wait states behind a
# buffer_store_dwordx3/x4 whose soffset is a register; round 5 saw such stores write out what a later vector instruction had
# put into their data registers.  Every one of them in the sweep kernels must keep its data registers untouched for
# ALQ_STORE_HOLD_STATES wait states - checked on the device assembly of THIS build, whatever the compiler or the flags did.
HZ=()
for f in $HAZARD_FILES; do
  A="$BUILD/$f-hip-amdgcn-amd-amdhsa-gfx950.s"
  if [ -f "$A" ]; then HZ+=("$A"); else echo "build.sh: device assembly of $f not found ($A): store-hazard gate cannot run" >&2; exit 1; fi
done
python3 "$ROOT/tools/isa_store_hazard.py" --min 2 "${HZ[@]}" > "$BUILD/store_hazard_report.txt" || { cat "$BUILD/store_hazard_report.txt" >&2; echo "build.sh: store-hazard gate failed" >&2; exit 1; }
for f in $HAZARD_FILES; do rm -f "$BUILD"/$f-hip-*.bc "$BUILD"/$f-hip-*.hipi "$BUILD"/$f-hip-*.s "$BUILD"/$f-hip-*.o "$BUILD"/$f-hip-*.out "$BUILD"/$f-hip-*.hipfb "$BUILD"/$f-host-*.bc "$BUILD"/$f-host-*.hipi "$BUILD"/$f-host-*.s; done
hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT" "$BUILD"/{igemm,igemm2,igemm3,igemm4,c3d,t3d,t3d8b,e3d,d3d,f3d,fcgemm,direct,kernels,topk,model,comm,train,sim,ref64}.o -ldl
echo "built $OUT"
