#!/bin/bash
# conv_duo_kernel (4-wave workgroups, two a CU) against conv_wide_kernel on one box: the kernel and model tests under
# CUTIE_CONV_DUO=1, then bench.py alternating.  tools/gpu_duo_ab.sh [tag] [rounds] ["levels"]   (CUTIE_CONV_DUO: 0 never, 1 the 64/128-channel tiles, 2 all)
set -o pipefail
tag=${1:-duo}; rounds=${2:-2}; levels=${3:-"1 0"}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
CUTIE_CONV_DUO=2 timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $O/pytest.log | head -30; exit $rc; }
for r in $(seq $rounds); do
  for v in $levels; do
    CUTIE_CONV_DUO=$v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_duo$v.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
    echo "duo=$v $(python -c "import json,sys; d=json.load(open('$O/bench_duo$v.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k: v['avg_launch_ms'] for k, v in d['roofline']['by_kind'].items()})")"
  done
done
