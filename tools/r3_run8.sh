cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3g
timeout 900 python3 tests/gpu_check.py bwdwide 2>&1 | grep -E "FAIL|SUMMARY" > gpurun_out/r3g/gpu_check.txt
for v in 0 1 0 1; do W2S_BWD_WIDE32=$v python3 bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('WIDE32=$v', d['ms_per_step'], d['value'], d['config']['final_loss'], {k:(v['ms'],v['GBps']) for k,v in d['roofline']['families'].items()})" >> gpurun_out/r3g/ab.txt 2>&1; cp gpurun_out/bench_launch_breakdown.json gpurun_out/r3g/breakdown_$v.json; done
W2S_BWD_WIDE32=1 W2S_BWD_WIDE32_WGS=768 python3 bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('WIDE32=1 wgs768', d['ms_per_step'], d['value'])" >> gpurun_out/r3g/ab.txt 2>&1
cat gpurun_out/r3g/gpu_check.txt gpurun_out/r3g/ab.txt
python3 - <<'PY'
import json
for v in (0,1):
    d=json.load(open(f'gpurun_out/r3g/breakdown_{v}.json'))
    print(v, {k:(x['launches'], round(x['ms'],3), round(x['bytes']/x['ms']/1e6)) for k,x in d.items() if 'bwd_wide' in k or 'bwd_fused_bf_kernel<2, 2' in k})
PY
