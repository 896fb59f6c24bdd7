#!/bin/bash
# launch census + timeline of the current tree, parity subset
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py tests/test_r3_parity_gpu.py tests/test_r4_parity_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.txt
bash tools/step_launches.sh > $O/launches.log 2>&1; cp gpurun_out/step_launches.txt $O/
bash tools/step_timeline.sh > $O/timeline.log 2>&1; cp gpurun_out/step_timeline.txt gpurun_out/step_kernels.txt $O/
for rep in 1 2; do timeout 600 python3 bench.py --no-cpu --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench rep $rep:', d['ms_per_step'], d['value'], d['config']['final_loss'])" >> $O/bench.txt; done
cat $O/pytest.txt; head -14 $O/step_launches.txt; cat $O/step_timeline.txt; cat $O/bench.txt
