#!/bin/bash
# A/B on one box: kernel tests, then bench with and without the fused norm-backward reduction, then kernel stats.
set -o pipefail
tag=${1:-ab}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $O/pytest.log | head -30; exit $rc; }
for v in 1 0 1 0; do
  CUTIE_FUSE_NORM_BWD=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f16 > $O/bench_fuse$v.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  echo "fuse=$v $(python -c "import json,sys; d=json.load(open('$O/bench_fuse$v.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])")"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f16 > $O/bench_prof.json 2> $O/prof.err || { tail -5 $O/prof.err; exit 1; }
cat $O/bench_prof.json
