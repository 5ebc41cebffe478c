#!/bin/bash
# A/B of the tree's library against variants/libcutie_head.so (the last commit, built by hand) on ONE box:
# kernel tests with the tree's library, single-layer timings with both, then bench.py alternating.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1 || exit 1
for lib in base head; do
  if [ $lib = base ]; then unset CUTIE_HIP_LIB; else export CUTIE_HIP_LIB=$R/variants/libcutie_$lib.so; fi
  for c in 64 256; do echo -n "$lib "; timeout -k 10 60 python tools/conv_bench.py fwd 1 50 $c 256 | tail -1 || exit 1; done
  echo -n "$lib "; timeout -k 10 60 python tools/dgrad_bench.py | tail -1 || exit 1
done
unset CUTIE_HIP_LIB
bash tools/gpu_variants.sh ${1:-2}
