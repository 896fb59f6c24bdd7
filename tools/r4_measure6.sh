#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
timeout 900 python3 tests/gpu_check.py fin fwdfused fusedbf fold first bwdwide wide wideup2 stats plumb 2>&1 | grep -E "FAIL|SUMMARY|Error|error" | head -40 > $O/gpu_check.txt
W2S_FUSED_FINALIZE=1 timeout 1500 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py -m gpu -q -x 2>&1 | tail -4 > $O/pytest_ff1.txt
for rep in 1 2 3; do
  for ff in 1 0; do
    W2S_FUSED_FINALIZE=$ff timeout 600 python3 bench.py --no-cpu --steps 15 2>$O/bench_ff$ff.$rep.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_finalize=$ff rep $rep:', d['ms_per_step'], d['value'], d['config']['final_loss'])" >> $O/bench_ab.txt 2>&1
  done
done
W2S_FUSED_FINALIZE=1 bash tools/step_launches.sh > $O/launches_ff1.log 2>&1; cp gpurun_out/step_launches.txt $O/step_launches_ff1.txt
cat $O/gpu_check.txt; cat $O/pytest_ff1.txt; cat $O/bench_ab.txt; head -4 $O/step_launches_ff1.txt
