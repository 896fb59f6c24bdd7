#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
# the round's closing run on the GPU box (tools/round_close.sh [tag]): stage checks, full suite, bench line, launch census, profile set
mkdir -p gpurun_out/r3z
timeout 900 python3 tests/gpu_check.py bwdwide gradh 2>&1 | grep -E "FAIL|SUMMARY" > gpurun_out/r3z/gpu_check.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3z/pytest.txt
python3 bench.py > gpurun_out/r3z/bench_line.json 2> gpurun_out/r3z/bench.err
bash tools/step_launches.sh > gpurun_out/r3z/census.log 2>&1; cp gpurun_out/step_launches.txt gpurun_out/r3z/step_launches.txt
bash tools/profile_bench.sh ${1:-r03} > gpurun_out/r3z/profile.log 2>&1
cat gpurun_out/r3z/gpu_check.txt gpurun_out/r3z/pytest.txt; cut -c1-600 gpurun_out/r3z/bench_line.json; head -5 gpurun_out/r3z/step_launches.txt; tail -3 gpurun_out/r3z/profile.log
