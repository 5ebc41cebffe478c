#!/bin/bash
# bench every library under variants/ against the in-tree one on ONE box: tools/gpu_variants.sh [rounds]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rounds=${1:-2}
for r in $(seq $rounds); do
  for lib in base $(ls variants/*.so 2>/dev/null); do
    if [ $lib = base ]; then unset CUTIE_HIP_LIB; else export CUTIE_HIP_LIB=$R/$lib; fi
    out=$(timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f16 2>/dev/null) || { echo "$lib FAILED"; exit 1; }
    echo "$lib $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['final_total_loss'])")"
  done
done
