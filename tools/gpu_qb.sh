#!/bin/bash
# quick tests + default bench + kernel profile: tools/gpu_qb.sh TAG pytest-args...
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest "$@" -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -4 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $O/pytest.log | head -40; exit $rc; }
timeout -k 10 240 python bench.py --no-cpu-baseline --no-alt --steps 40 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['by_kind'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-kernel-timing --no-clock > $O/bench_prof.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_prof.json')); print('profiled', d['value'], d['ms_per_step'])"
python $R/tools/prof_summary.py $O/prof/*/*_kernel_stats.csv 33 60 | cut -c1-70,93-130
