#!/bin/bash
# per-launch conv times of a step, one table per setting of an env switch: tools/gpu_launch_table.sh TAG VAR
set -o pipefail
tag=$1; var=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
for v in 1 0; do
  env $var=$v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt --launch-table > $O/bench_$v.json 2> $O/table_$v.txt || { tail -5 $O/table_$v.txt; exit 1; }
done
paste -d'|' <(cut -c1-80 $O/table_1.txt) <(cut -c70-90 $O/table_0.txt)
