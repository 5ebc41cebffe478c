#!/bin/bash
# conv_bench over a few layer shapes for every library under variants/ against the shipped one: tools/gpu_diag_shapes.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for r in 1 2; do
for lib in base $(ls variants/*.so 2>/dev/null); do
  if [ $lib = base ]; then unset CUTIE_HIP_LIB; else export CUTIE_HIP_LIB=$R/$lib; fi
  for shape in "256 256" "512 64" "64 256" "128 256"; do
    echo "$lib $(timeout -k 10 120 python tools/conv_bench.py fwd 1 40 $shape 2>&1 | tail -1)"
  done
done
done
