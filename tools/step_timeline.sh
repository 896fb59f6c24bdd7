#!/bin/bash
# Phase / stream timeline of ONE steady-state train step as bench.py runs it (multi-stream): tools/step_timeline.sh -> gpurun_out/step_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl   # (a second run in one session must not pick up the first run's trace)
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob, collections
kt = glob.glob('gpurun_out/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adamw_kernel')]
t_prev_end = int(rows[ends[-2]]['End_Timestamp'])
step = [r for r in rows if int(r['Start_Timestamp']) >= t_prev_end and int(r['Start_Timestamp']) <= int(rows[ends[-1]]['Start_Timestamp'])]
T0 = t_prev_end
out = [f'step wall {(int(step[-1]["End_Timestamp"]) - T0) / 1e6:.3f} ms, {len(step)} kernels']
byq = collections.defaultdict(list)
for r in step: byq[r['Queue_Id']].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: int(kv[1][0]['Start_Timestamp'])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs) / 1e6
    # contiguous activity segments (gap > 0.3 ms splits)
    segs = []; s0 = int(rs[0]['Start_Timestamp']); e = int(rs[0]['End_Timestamp'])
    for r in rs[1:]:
        if int(r['Start_Timestamp']) - e > 300000: segs.append((s0, e)); s0 = int(r['Start_Timestamp'])
        e = max(e, int(r['End_Timestamp']))
    segs.append((s0, e))
    out.append(f'queue {q}: {len(rs)} kernels, busy {busy:.2f} ms, segments ' + ' '.join(f'[{(a - T0) / 1e6:.2f}-{(b - T0) / 1e6:.2f}]' for a, b in segs))
# union busy / idle
ev = sorted([(int(r['Start_Timestamp']), 1) for r in step] + [(int(r['End_Timestamp']), -1) for r in step])
act = 0; last = T0; idle = 0; gaps = []
for t, d in ev:
    if act == 0 and t > last:
        idle += t - last
        if t - last > 20000: gaps.append(((last - T0) / 1e6, (t - last) / 1e3))
    act += d; last = t if act == 0 else last
    if act == 0: last = t
out.append(f'device idle within the step: {idle / 1e6:.3f} ms; gaps > 20 us: ' + ', '.join(f'{a:.2f}ms:{g:.0f}us' for a, g in gaps[:30]))
with open('gpurun_out/step_kernels.txt', 'w') as fk:   # every launch of the step: start (us), duration (us), queue, name
    for r in step:
        fk.write(f"{(int(r['Start_Timestamp']) - T0) / 1e3:10.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} q{r['Queue_Id']} {r['Kernel_Name'][:110]}\n")
open('gpurun_out/step_timeline.txt', 'w').write('\n'.join(out) + '\n'); print('\n'.join(out))
PY
