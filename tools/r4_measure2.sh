#!/bin/bash
# Round-4 second pass on the GPU box: the per-workgroup in-kernel statistics finalisation (blocked tile assignment) -- kernel stage checks,
# isolated kernels against the round-3 library (grid-stride assignment: build_alt/libw2s_r3.so), parity tests, and the step with / without
# the finalize launches.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
timeout 1200 python3 tests/gpu_check.py fin fwdfused fusedbf fold first bwdwide wide wideup2 stats conv 2>&1 | grep -E "FAIL|SUMMARY|Error|error" | head -60 > $O/gpu_check.txt
CASES="ff16 ff16s2 ff1632 ff32 ff32s2 ffirst b16 b16u b32 b32u b21 f64 f64s2 f128 f128s2 d64 d128 u64 u128"
for rep in 1 2; do
  BF=1 timeout 600 python3 tools/kbench.py $CASES --iters 20 > $O/kbench_new.$rep.txt 2>&1
  BF=1 W2S_LIB=$PWD/build_alt/libw2s_r3.so timeout 600 python3 tools/kbench.py $CASES --iters 20 > $O/kbench_r3.$rep.txt 2>&1
done
paste <(grep us $O/kbench_new.1.txt | awk '{print $1, $2}') <(grep us $O/kbench_r3.1.txt | awk '{print $2}') <(grep us $O/kbench_new.2.txt | awk '{print $2}') <(grep us $O/kbench_r3.2.txt | awk '{print $2}') > $O/kbench_ab.txt
timeout 2400 python3 -m pytest tests/test_r4_parity_gpu.py tests/test_parity_gpu.py -m gpu -q -x 2>&1 | tail -25 > $O/pytest.txt
for rep in 1 2; do
  W2S_FUSED_FINALIZE=1 W2S_LIB=$PWD/build_alt/libw2s_nofence.so timeout 600 python3 bench.py --no-cpu --no-extra --steps 15 2>$O/bench_nf.$rep.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_finalize=1 no acquire fence rep $rep:', d['ms_per_step'], d['value'], d['config']['final_loss'])" >> $O/bench_ab.txt 2>&1
  for ff in 1 0; do
    W2S_FUSED_FINALIZE=$ff timeout 600 python3 bench.py --no-cpu --no-extra --steps 15 2>$O/bench_ff$ff.$rep.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_finalize=$ff rep $rep:', d['ms_per_step'], d['value'], d['config']['final_loss'])" >> $O/bench_ab.txt 2>&1
  done
done
cat $O/gpu_check.txt; echo; echo "case new r3 new r3 (us)"; cat $O/kbench_ab.txt; cat $O/pytest.txt | tail -8; cat $O/bench_ab.txt
