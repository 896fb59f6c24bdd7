#!/bin/bash
# usage: tools/pmc.sh <outdir> <kbench case>     -- separate --pmc passes (no trace domains mixed in)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$1; CASE=$2
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  if [ -n "$PMC_PASSES" ] && [ $i -gt $PMC_PASSES ]; then break; fi
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 tools/kbench.py $CASE --iters 3 > $OUT.p$i.log 2>&1 || true
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'conv_cl' in k or 'wgrad_' in k or 'bwd_fused' in k or 'conv_wide' in k or 'conv_fwd' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print(f'   {c:28s} mean/dispatch {sum(v)/len(v):16.1f}  (n={len(v)})')
PY
