#!/bin/bash
# Launch census of ONE steady-state train step (run on the GPU box): tools/step_launches.sh  -> gpurun_out/step_launches.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export W2S_MULTI_STREAM=0
rm -rf gpurun_out/steptrace   # (a second run in one session must not pick up the first run's trace)
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/steptrace -- python3 bench.py --steps 3 --warmup 2 --no-cpu --no-roofline > gpurun_out/steptrace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
kt = glob.glob('gpurun_out/steptrace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adamw_kernel')]
lo, hi = ends[-2] + 1, ends[-1] + 1     # the last full step
step = rows[lo:hi]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
cnt = collections.Counter(); dur = collections.Counter()
for r in step:
    k = r['Kernel_Name'].split('(')[0][:90]; cnt[k] += 1; dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
out = [f'kernel launches in one step: {len(step)}; wall {(t1 - t0) / 1e6:.3f} ms; sum of kernel durations {sum(dur.values()) / 1e6:.3f} ms']
mc = glob.glob('gpurun_out/steptrace/**/*memory_copy_trace.csv', recursive=True)
if mc:
    cps = [r for r in csv.DictReader(open(mc[0])) if t0 <= int(r['Start_Timestamp']) <= t1]
    out.append(f'memory copies in that step: {len(cps)}: ' + str(collections.Counter(r.get('Direction', '?') for r in cps)))
for k, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    out.append(f'{n:5d} {dur[k] / 1e3:9.1f} us  {k}')
open('gpurun_out/step_launches.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[:12]))
PY
