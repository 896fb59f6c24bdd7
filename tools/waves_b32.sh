#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for w in 1 2 1 2; do
W2S_WAVES=$w python3 bench.py --batch 32 --steps 6 --warmup 2 --no-cpu --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $w batch 32:', d['ms_per_step'], d['value'])"
done
