"""Kernel-trace gap analysis: python tools/gap_analysis.py <kernel_trace.csv> -- busy time, idle gaps and launch count per train step."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# steps are delimited by the adamw kernel
ends = [e for s, e, n in ev if n.startswith('adamw_kernel')]
prev = None
for k, cut in enumerate(ends):
    step = [(s, e, n) for s, e, n in ev if (prev is None or s >= prev) and e <= cut]
    prev = cut
    if not step:
        continue
    t0, t1 = step[0][0], max(e for _, e, _ in step)
    busy = 0; cur_s, cur_e = step[0][0], step[0][1]
    for s, e, _ in step[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in step)
    small = sum(1 for s, e, _ in step if e - s < 10000)
    print(f'step {k}: launches {len(step)}, wall {1e-6*(t1-t0):.2f} ms, GPU busy (union) {1e-6*busy:.2f} ms, idle {1e-6*(t1-t0-busy):.2f} ms, sum of kernel durations {1e-6*ksum:.2f} ms, launches < 10 us: {small}')
