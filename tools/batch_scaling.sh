#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
export W2S_MULTI_STREAM=0
for b in 8 16; do
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b$b -- python3 bench.py --steps 4 --warmup 2 --batch $b --no-cpu --no-roofline > gpurun_out/b$b.log 2>&1
cp $(ls gpurun_out/b$b/*/*kernel_stats.csv | head -1) gpurun_out/b${b}_stats.csv
tail -1 gpurun_out/b$b.log | cut -c1-200
done
