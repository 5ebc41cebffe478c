#!/bin/bash
# A/B of a variant library against the shipped one on one box: tools/gpu_lib_ab.sh TAG VARIANT_NAME
set -o pipefail
tag=$1; var=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
CUTIE_HIP_LIB=$R/variants/libcutie_$var.so timeout -k 10 400 python -u -m pytest tests/test_kernels_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "conv" > $O/pytest.log 2>&1
rc=$?; tail -2 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $O/pytest.log | head -30; exit $rc; }
for v in $var base $var base; do
  if [ $v = base ]; then unset CUTIE_HIP_LIB; else export CUTIE_HIP_LIB=$R/variants/libcutie_$v.so; fi
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_$v.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  echo "$v $(python -c "import json,sys; d=json.load(open('$O/bench_$v.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k: v['avg_launch_ms'] for k, v in d['roofline']['by_kind'].items()})")"
done
