cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3i
timeout 900 python3 tests/gpu_check.py bwdwide 2>&1 | grep -E "FAIL|SUMMARY" > gpurun_out/r3i/gpu_check.txt
timeout 1800 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py tests/test_r3_parity_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3i/pytest.txt
bash tools/ab_bench.sh "base:W2S_BWD_WIDE_RD=0" "rd:W2S_BWD_WIDE_RD=1" "base:W2S_BWD_WIDE_RD=0" "rd:W2S_BWD_WIDE_RD=1" 2>&1 | tail -10 > gpurun_out/r3i/ab.txt
cat gpurun_out/r3i/*.txt
