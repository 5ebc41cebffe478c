#!/bin/bash
# one-rank rehearsal of the multi-GPU step on a single GPU: RCCL group of one, bucketed all-reduce path forced
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python -u -m pytest tests/test_dp_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
for v in plain buckets plain buckets; do
  if [ $v = buckets ]; then export CUTIE_DP_FORCE_BUCKETS=1; else unset CUTIE_DP_FORCE_BUCKETS; fi
  out=$(timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-f16 --no-kernel-timing 2>/dev/null | grep '^{') || { echo "$v FAILED"; exit 1; }
  echo "$v $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['final_total_loss'])")"
done
