cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3b
timeout 300 build_alt/pk_repro > gpurun_out/r3b/pk_repro2.txt 2>&1
for dbg in 0 2 1 4 8 16 6 7; do echo "W2S_WIDE_DBG=$dbg" >> gpurun_out/r3b/race_bisect.txt; W2S_WIDE_DBG=$dbg MODES=conv_wide RUNS=20 W2S_LIB=$PWD/build_alt/libw2s_f4.so timeout 300 python3 tools/first_bwd_race.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3b/race_bisect.txt; done
timeout 900 python3 tests/gpu_check.py gradh fusedbf fold first > gpurun_out/r3b/gpu_check.txt 2>&1
timeout 1500 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py tests/test_r3_parity_gpu.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3b/pytest.txt
tools/hang_hunt.sh 1 eog_fullsize_grad b16_fullsize_grad > gpurun_out/r3b/fullsize.txt 2>&1
for v in 0 1 0 1; do W2S_GRAD_FP16=$v python3 bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('W2S_GRAD_FP16=$v', d['ms_per_step'], d['value'], d['config']['final_loss'], {k:(v['ms'],v['GBps']) for k,v in d['roofline']['families'].items()})" >> gpurun_out/r3b/ab.txt 2>&1; done
tail -n 40 gpurun_out/r3b/*.txt
