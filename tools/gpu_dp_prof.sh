#!/bin/bash
# kernel profile of the one-rank rehearsal of the data-parallel step (RCCL group of one, bucketed path forced)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-dpprof}
mkdir -p $O
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 CUTIE_DP_FORCE_BUCKETS=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-kernel-timing --no-clock > $O/bench.json 2> $O/err.log || { tail -5 $O/err.log; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print('profiled dp', d['value'], d['ms_per_step'])"
python $R/tools/prof_summary.py $O/prof/*/*_kernel_stats.csv 33 70 | cut -c1-70,93-130
