#!/bin/bash
# A/B of one library switch on ONE box (boxes differ by up to 12 %): tools/gpu_env_ab.sh VAR [tag] [rounds]
# kernel + model tests first (with the default), then bench.py alternating VAR=1 / VAR=0.
set -o pipefail
var=$1; tag=${2:-envab}; rounds=${3:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_full_size_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $O/pytest.log | head -30; exit $rc; }
for r in $(seq $rounds); do
  for v in 1 0; do
    env $var=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_$v.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
    echo "$var=$v $(python -c "import json,sys; d=json.load(open('$O/bench_$v.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['final_total_loss'])")"
  done
done
