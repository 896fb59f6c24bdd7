#!/bin/bash
# A/B of the low-degree erf build (build_alt/libw2s_lowdeg.so: -DW2S_ERF_LOWDEG=1): isolated kernels, parity tests, the step
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
L=$PWD/build_alt/libw2s_lowdeg.so
CASES="ff16 ff16s2 ff32 ffirst b16 b16u b32 b32u bfirst d64 d128"
for rep in 1 2; do
  BF=1 timeout 600 python3 tools/kbench.py $CASES --iters 20 > $O/kbench_base.$rep.txt 2>&1
  BF=1 W2S_LIB=$L timeout 600 python3 tools/kbench.py $CASES --iters 20 > $O/kbench_low.$rep.txt 2>&1
done
paste <(grep us $O/kbench_base.1.txt | awk '{print $1, $2}') <(grep us $O/kbench_low.1.txt | awk '{print $2}') <(grep us $O/kbench_base.2.txt | awk '{print $2}') <(grep us $O/kbench_low.2.txt | awk '{print $2}') > $O/kbench_ab.txt
W2S_LIB=$L timeout 2400 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest_low.txt
W2S_LIB=$L timeout 900 python3 tests/child_checks.py b16_fullsize_grad $O/b16_low.json > $O/b16_low.txt 2>&1
for rep in 1 2; do
  for lib in base low; do
    LIBENV=""; [ $lib = low ] && LIBENV="W2S_LIB=$L"
    env $LIBENV timeout 600 python3 bench.py --no-extra --steps 15 --no-cpu 2>$O/bench_$lib.$rep.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep:', d['ms_per_step'], d['value'], d['config']['final_loss'], d.get('kappa_parity'))" >> $O/bench_ab.txt 2>&1
  done
done
echo "case base low base low (us)"; cat $O/kbench_ab.txt; cat $O/pytest_low.txt; tail -3 $O/b16_low.txt; cat $O/bench_ab.txt
