cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3h
for l in mt11 mt11b; do W2S_LIB=$PWD/build_alt/libw2s_$l.so timeout 900 python3 tests/gpu_check.py fusedbf first fold gradh 2>&1 | grep -E "FAIL|SUMMARY" > gpurun_out/r3h/gpu_check_$l.txt; done
W2S_LIB=$PWD/build_alt/libw2s_mt11.so W2S_BWD_WGS=768 timeout 1800 python3 -m pytest tests/test_parity_gpu.py tests/test_r2_parity_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3h/pytest_mt11.txt
bash tools/ab_bench.sh "base" "mt11:W2S_BWD_WGS=768" "mt11b:W2S_BWD_WGS=1024" "base" 2>&1 | tail -10 > gpurun_out/r3h/ab.txt
cat gpurun_out/r3h/*.txt
