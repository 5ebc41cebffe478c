#!/bin/bash
# A/B builds: tools/build_variant.sh NAME [-DFLAG ...]  ->  variants/libcutie_NAME.so (select with CUTIE_HIP_LIB)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/variants/obj_$name
pids=()
for f in core grid_build embed norm loss optim conv_mfma conv_wide; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" \
    -c $R/cutie_amd/csrc/$f.hip -o $R/variants/obj_$name/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/libcutie_$name.so $R/variants/obj_$name/*.o
rm -rf $R/variants/obj_$name
echo built $R/variants/libcutie_$name.so
