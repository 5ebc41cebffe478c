#!/bin/bash
# One GPU-box pass: GPU tests, default bench, rocprof kernel stats of the bench.  Output under gpurun_out/$1.
set -o pipefail
tag=${1:-chk}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 240 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-clock > $O/bench_prof.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
cat $O/bench_prof.json
