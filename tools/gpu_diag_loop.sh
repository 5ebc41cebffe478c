#!/bin/bash
# what the conv_wide K loop waits for: a dense 256 -> 256 3x5 forward launch on every diagnostic library under variants/
# (tools/build_variant.sh NAME -DCW_DIAG_...; wrong results by design) against the shipped one.  tools/gpu_diag_loop.sh [dil]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
dil=${1:-1}
for r in 1 2; do
for lib in base $(ls variants/*.so 2>/dev/null); do
  if [ $lib = base ]; then unset CUTIE_HIP_LIB; else export CUTIE_HIP_LIB=$R/$lib; fi
  echo "$lib $(timeout -k 10 120 python tools/conv_bench.py fwd $dil 40 2>&1 | tail -1)"
done
done
