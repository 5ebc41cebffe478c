#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -u -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
for v in 1 0 1 0; do
  out=$(CUTIE_OVERLAP_WGRAD=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f16 --no-kernel-timing 2>/dev/null) || { echo FAILED; exit 1; }
  echo "overlap=$v $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_total_loss'])")"
done
