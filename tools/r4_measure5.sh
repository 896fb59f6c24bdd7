#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
timeout 1200 python3 tests/gpu_check.py fusedbf fold gradh first fin plumb 2>&1 | grep -E "FAIL|SUMMARY|Error|error" | head -40 > $O/gpu_check.txt
CASES="b16 b16u b32 b32u bfirst ff16 ff32"
for rep in 1 2; do
  BF=1 timeout 600 python3 tools/kbench.py $CASES --iters 20 > $O/kbench_new.$rep.txt 2>&1
  BF=1 W2S_LIB=$PWD/build_alt/libw2s_r3.so timeout 600 python3 tools/kbench.py b16 b16u b32 b32u ff16 ff32 --iters 20 > $O/kbench_r3.$rep.txt 2>&1
done
timeout 2400 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
timeout 900 python3 tests/child_checks.py b16_fullsize_grad $O/b16.json > $O/b16.txt 2>&1
timeout 900 python3 tests/child_checks.py eog_fullsize_grad $O/eog.json > $O/eog.txt 2>&1
for rep in 1 2; do timeout 600 python3 bench.py --no-cpu --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench rep $rep:', d['ms_per_step'], d['value'], d['config']['final_loss'])" >> $O/bench.txt; done
cat $O/gpu_check.txt; grep us $O/kbench_new.1.txt; echo r3; grep us $O/kbench_r3.1.txt; echo rep2; grep us $O/kbench_new.2.txt; echo r3; grep us $O/kbench_r3.2.txt; cat $O/pytest.txt; cat $O/b16.json; echo; cat $O/eog.json; echo; cat $O/bench.txt
