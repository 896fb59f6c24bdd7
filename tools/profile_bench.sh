#!/bin/bash
# Round profile of the headline benchmark (run on the GPU box):  tools/profile_bench.sh <tag>
#   1. rocprofv3 --kernel-trace --stats      -> gpurun_out/<tag>_stats/
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no trace domains) -> per-kernel HBM traffic
#   3. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -> per-kernel matrix-pipe utilisation
# Summaries are written to gpurun_out/<tag>_kernel_stats.csv and gpurun_out/<tag>_pmc_traffic.json (copy to profiles/).
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- $CMD > gpurun_out/${TAG}_stats.log 2>&1
export W2S_MULTI_STREAM=0   # isolated per-kernel durations / traffic: one stream (the roofline leg of bench.py runs like this)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats1s -- $CMD > gpurun_out/${TAG}_stats1s.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_fetch -- $CMD > gpurun_out/${TAG}_fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_write -- $CMD > gpurun_out/${TAG}_write.log 2>&1 || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_mfma -- $CMD > gpurun_out/${TAG}_mfma.log 2>&1 || true
python3 - <<PY
import csv, glob, json, collections
tag = '$TAG'
st = glob.glob(f'gpurun_out/{tag}_stats/**/*kernel_stats.csv', recursive=True)[0]
open(f'gpurun_out/{tag}_kernel_stats.csv', 'w').write('# rocprofv3 --kernel-trace --stats --output-format csv -- $CMD  (3 train steps, batch 16, MI355X; encoders on 4 HIP streams)\n' + open(st).read())
st1 = glob.glob(f'gpurun_out/{tag}_stats1s/**/*kernel_stats.csv', recursive=True)[0]
open(f'gpurun_out/{tag}_kernel_stats_single_stream.csv', 'w').write('# W2S_MULTI_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -- $CMD  (3 train steps, batch 16, MI355X; one stream)\n' + open(st1).read())
agg = collections.defaultdict(lambda: {'FETCH_SIZE': [], 'WRITE_SIZE': []})
for kind in ('fetch', 'write'):
    for f in glob.glob(f'gpurun_out/{tag}_{kind}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in agg.items():
    if not d['FETCH_SIZE'] or not d['WRITE_SIZE']:
        continue
    fetch = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']) * 1024 * 2   # KB -> B; gfx950: FETCH_SIZE reports 1/2 of a wide coalesced stream (MI355X_MICROARCH.md, HBM)
    write = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE']) * 1024
    name = k.split('(')[0].replace('void ', '')
    out[name] = {'launches': len(d['FETCH_SIZE']), 'read_bytes_per_launch': fetch, 'write_bytes_per_launch': write, 'hbm_bytes_per_launch': fetch + write}
json.dump(out, open(f'gpurun_out/{tag}_pmc_traffic.json', 'w'), indent=1)
print('kernels with traffic:', len(out))
# matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (16 per v_mfma_f32_16x16x32_bf16,
# MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so kernel cycles = GRBM_GUI_ACTIVE / 8
m = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/{tag}_mfma/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
mo = {}
for k, d in m.items():
    if not d['SQ_VALU_MFMA_BUSY_CYCLES'] or not d['GRBM_GUI_ACTIVE'] or sum(d['SQ_INSTS_MFMA']) == 0:
        continue
    n = len(d['GRBM_GUI_ACTIVE'])
    busy, cyc, insts = sum(d['SQ_VALU_MFMA_BUSY_CYCLES']) / n, sum(d['GRBM_GUI_ACTIVE']) / n / 8, sum(d['SQ_INSTS_MFMA']) / n
    name = k.split('(')[0].replace('void ', '')
    mo[name] = {'launches': n, 'mfma_busy_cycles_per_launch': busy, 'mfma_insts_per_launch': insts, 'kernel_cycles_per_launch': cyc,
                'mfma_pipe_utilisation': busy / (cyc * 1024) if cyc else None}
json.dump(mo, open(f'gpurun_out/{tag}_pmc_mfma.json', 'w'), indent=1)
print('kernels with MFMA counters:', len(mo))
PY
grep metric gpurun_out/${TAG}_stats.log | cut -c1-200
