#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export W2S_FORCE_COLLECTIVES=1
rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d gpurun_out/coll -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-roofline > gpurun_out/coll.log 2>&1
grep metric gpurun_out/coll.log | cut -c60-170
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/coll/**/*hip_api_stats.csv', recursive=True) + glob.glob('gpurun_out/coll/**/*hip_stats.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:14]:
        print(r['Calls'], f"{float(r['TotalDurationNs'])/1e6:.1f} ms", f"{float(r['AverageNs'])/1e3:.1f} us", r['Name'])
PY
ls gpurun_out/coll/*/ | head
