#!/bin/bash
# usage: tools/ab_kbench.sh "<kbench cases>" name1 [name2 ...]   -- isolated kernels (tools/kbench.py, BF=1) under the in-tree library ('base')
# and alternative libraries build_alt/libw2s_<name>.so (tools/altlib.sh), two alternating repetitions each -> gpurun_out/abk/<name>.<rep>.txt
# and a side-by-side table.  Round 4 used it for: the erf-free skeleton build, the low-degree erf, blocked vs grid-stride tile assignment.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
CASES=$1; shift
mkdir -p gpurun_out/abk
for rep in 1 2; do
  for name in "$@"; do
    LIBENV=""; [ "$name" != base ] && LIBENV="W2S_LIB=$PWD/build_alt/libw2s_$name.so"
    env BF=1 $LIBENV timeout 900 python3 tools/kbench.py $CASES --iters 20 > gpurun_out/abk/$name.$rep.txt 2>&1
  done
done
first=$1
paste <(grep ' us ' gpurun_out/abk/$first.1.txt | awk '{print $1}') $(for name in "$@"; do for rep in 1 2; do echo "<(grep ' us ' gpurun_out/abk/$name.$rep.txt | awk '{print \$2}')"; done; done | tr '\n' ' ') 2>/dev/null || true
echo "columns: case, then (rep 1, rep 2) microseconds for each of: $*"
for name in "$@"; do for rep in 1 2; do echo "== $name rep $rep"; grep ' us ' gpurun_out/abk/$name.$rep.txt; done; done
