cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3f
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3f/pytest.txt
for v in 1 2 3; do python3 bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['final_loss'], {k:(v['ms'],v['GBps']) for k,v in d['roofline']['families'].items()})" >> gpurun_out/r3f/ab.txt 2>&1; done
bash tools/step_launches.sh > gpurun_out/r3f/census.log 2>&1; cp gpurun_out/step_launches.txt gpurun_out/r3f/step_launches.txt
cat gpurun_out/r3f/pytest.txt; cat gpurun_out/r3f/ab.txt; head -12 gpurun_out/r3f/step_launches.txt | cut -c1-150; grep -E "eltwise|enc_first_bwd|conv_cl_kernel<8, 4, 1, 1|conv_cl_kernel<8, 4, 4, 4" gpurun_out/r3f/step_launches.txt
