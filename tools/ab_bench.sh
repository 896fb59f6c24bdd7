#!/bin/bash
# usage: tools/ab_bench.sh "name1:ENV=.. ENV2=.." "name2:..."   -- bench.py (no CPU leg) per configuration; name 'base' = in-tree library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  [ "$name" = "$spec" ] && envs=""
  lib=""; [ -f build_alt/libw2s_$name.so ] && lib="W2S_LIB=$GRAFT_REPO_ROOT/build_alt/libw2s_$name.so"
  for rep in 1 2; do
    env $lib $envs python3 bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/ab/$name.$rep.json 2> gpurun_out/ab/$name.$rep.err
    python3 - <<PY
import json
try:
    d=json.loads(open('gpurun_out/ab/$name.$rep.json').read().strip().splitlines()[-1]); print('$name', '$rep', d['ms_per_step'], d['value'], d['config']['final_loss'])
except Exception as e: print('$name FAILED', e); print(open('gpurun_out/ab/$name.$rep.err').read()[-1500:])
PY
  done
  cp gpurun_out/bench_launch_breakdown.json gpurun_out/ab/$name.breakdown.json 2>/dev/null
done
