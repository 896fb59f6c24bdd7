cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s
timeout 1500 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py tests/test_r3_parity_gpu.py tests/test_r2_serving_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3s/pytest.txt
bash tools/ab_bench.sh "base:W2S_SEQ_SPLIT=1" "split:W2S_SEQ_SPLIT=2" "base:W2S_SEQ_SPLIT=1" "split:W2S_SEQ_SPLIT=2" "split4:W2S_SEQ_SPLIT=4" 2>&1 | tail -12 > gpurun_out/r3s/ab.txt
cat gpurun_out/r3s/pytest.txt gpurun_out/r3s/ab.txt
