#!/bin/bash
# Forward-only (entropy filter) throughput of NET-C at 32^3 under two environments: 200k-patch filter of the AL loop, 2 rounds each.
#   tools/ab_filter.sh "ENVA=1" "-"
set -eo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
cd "$ROOT"; mkdir -p gpurun_out
for tag in a b a b; do
  E="$1"; [ "$tag" = b ] && E="$2"; [ "$E" = "-" ] && E=""
  ( for kv in $E; do export "$kv"; done
    python3 -c "import sys; sys.path.insert(0,'.'); import nnal_amd; from nnal_amd import al_loop; al_loop.main()" 200000 2 2>&1 | grep "^round" | sed "s/^/$tag: /" | cut -c1-90 )
done
