#!/bin/bash
# Round-4 first measurement pass on the GPU box: the tuned stream ubench, the new / re-wired parity tests,
# the bench line with its `extra` legs, the <= 32-channel kernels with and without their transcendental work (build_alt/libw2s_noerf.so =
# tools/altlib.sh noerf "-DW2S_ERF_IDENTITY=1" bwd_fused.hip fwd_fused.hip), and the SQ issue counters of the five largest kernels.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
{ rocminfo 2>/dev/null | grep -m1 -i "marketing name"; cat /opt/rocm/.info/version 2>/dev/null; uname -r; } > $O/box.txt 2>&1
timeout 300 build_alt/stream_tuned > $O/stream.txt 2>&1
timeout 1500 python3 -m pytest tests/test_r4_parity_gpu.py "tests/test_parity_gpu.py::test_kernels_against_cpu_torch" -m gpu -q -x -k "r4 or bwdwide or wideup2 or wgwide" 2>&1 | tail -15 > $O/pytest.txt
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench.err
cp gpurun_out/bench_launch_breakdown.json $O/ 2>/dev/null
CASES="ff16 ff16s2 ff1632 ff32 ff32s2 ffirst b16 b16u b32 b32u b21 bfirst"
BF=1 timeout 600 python3 tools/kbench.py $CASES --iters 20 > $O/kbench_base.txt 2>&1
BF=1 W2S_LIB=$PWD/build_alt/libw2s_noerf.so timeout 600 python3 tools/kbench.py $CASES --iters 20 > $O/kbench_noerf.txt 2>&1
for c in b16u bfirst ffirst b32u ff16; do
  BF=1 PMC_PASSES=2 timeout 900 bash tools/pmc.sh $O/pmc_$c $c > $O/pmc_$c.txt 2>&1
done
tail -n 40 $O/stream.txt $O/pytest.txt $O/kbench_base.txt $O/kbench_noerf.txt
cat $O/bench_line.json | cut -c1-1500
