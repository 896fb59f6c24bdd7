cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3e
timeout 900 python3 tests/gpu_check.py bwdwide > gpurun_out/r3e/gpu_check.txt 2>&1
timeout 1800 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py tests/test_r3_parity_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3e/pytest.txt
bash tools/step_launches.sh > gpurun_out/r3e/census.log 2>&1; cp gpurun_out/step_launches.txt gpurun_out/r3e/step_launches.txt
for v in 1 1; do python3 bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['final_loss'], {k:(v['ms'],v['GBps']) for k,v in d['roofline']['families'].items()})" >> gpurun_out/r3e/ab.txt 2>&1; done
grep -E "FAIL|SUMMARY" gpurun_out/r3e/gpu_check.txt; cat gpurun_out/r3e/pytest.txt; cat gpurun_out/r3e/ab.txt; head -70 gpurun_out/r3e/step_launches.txt | cut -c1-150
