#!/bin/bash
# One GPU-box pass over named test files (or -k expressions): tools/gpu_quick.sh TAG pytest-args...
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest "$@" -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -4 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $O/pytest.log | head -40; }
exit $rc
