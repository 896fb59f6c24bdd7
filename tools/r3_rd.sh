cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3j
timeout 900 python3 tests/gpu_check.py bwdwide 2>&1 | grep -E "FAIL|SUMMARY" > gpurun_out/r3j/gpu_check.txt
bash tools/ab_bench.sh "base:W2S_BWD_WIDE_RD32=0" "rd:W2S_BWD_WIDE_RD32=1" "base:W2S_BWD_WIDE_RD32=0" "rd:W2S_BWD_WIDE_RD32=1" 2>&1 | tail -10 > gpurun_out/r3j/ab.txt
W2S_BWD_WIDE_RD32=1 timeout 1200 python3 -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "train_steps or autograd or full_size" 2>&1 | tail -3 > gpurun_out/r3j/pytest.txt
cat gpurun_out/r3j/*.txt
