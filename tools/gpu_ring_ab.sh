#!/bin/bash
# conv_wide K loop A/B on one box: kernel tests with the ring, then bench ring / two-stage alternating
set -o pipefail
tag=${1:-ring}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $O/pytest.log | head -30; exit $rc; }
for v in 1 0 1 0; do
  CUTIE_CONV_RING=$v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_ring$v.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  echo "ring=$v $(python -c "import json,sys; d=json.load(open('$O/bench_ring$v.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['by_kind'])")"
done
