cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3d
timeout 900 python3 tests/gpu_check.py bwdwide first > gpurun_out/r3d/gpu_check.txt 2>&1
timeout 1800 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py tests/test_r3_parity_gpu.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3d/pytest.txt
tools/hang_hunt.sh 1 b16_fullsize_grad eog_fullsize_grad > gpurun_out/r3d/fullsize.txt 2>&1
for v in 1 0 1; do W2S_BWD_WIDE=$v python3 bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('W2S_BWD_WIDE=$v', d['ms_per_step'], d['value'], d['config']['final_loss'], {k:(v['ms'],v['GBps']) for k,v in d['roofline']['families'].items()})" >> gpurun_out/r3d/ab.txt 2>&1; cp gpurun_out/bench_launch_breakdown.json gpurun_out/r3d/breakdown_$v.json; done
grep -c OK gpurun_out/r3d/gpu_check.txt; grep -E "FAIL|SUMMARY" gpurun_out/r3d/gpu_check.txt; tail -n 12 gpurun_out/r3d/pytest.txt; cat gpurun_out/r3d/fullsize.txt | cut -c1-300; cat gpurun_out/r3d/ab.txt
