#!/bin/bash
# the whole GPU suite, then the default bench: tools/gpu_full.sh TAG
set -o pipefail
tag=${1:-full}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -3 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $O/pytest.log | head -40; exit $rc; }
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['config']['whole_step_mfma_frac'], d['roofline']['frac'], d.get('bf16_path'), d['cpu_baseline'])"
