#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/r3s
timeout 900 python3 tests/gpu_check.py conv stats dgrad 2>&1 | grep -E "FAIL|SUMMARY" > gpurun_out/r3s/gpu_check.txt
timeout 1500 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3s/pytest.txt
bash tools/ab_bench.sh "base:W2S_DS_CONTIG=1" "ds" "base:W2S_DS_CONTIG=1" "ds" 2>&1 | tail -10 > gpurun_out/r3s/ab.txt
cat gpurun_out/r3s/gpu_check.txt gpurun_out/r3s/pytest.txt gpurun_out/r3s/ab.txt
python3 - <<'PY'
import json
for n in ('base','ds'):
    d=json.load(open(f'gpurun_out/ab/{n}.breakdown.json'))
    print(n, {k:(x['launches'], round(x['ms'],3)) for k,x in d.items() if 'conv_cl_kernel' in k and ', 1, 2, ' in k})
PY
