cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3k
W2S_GRAD_FP16=1 timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3k/pytest_fp16.txt
W2S_GRAD_FP16=1 tools/hang_hunt.sh 1 eog_fullsize_grad b16_fullsize_grad > gpurun_out/r3k/fullsize_fp16.txt 2>&1
bash tools/ab_bench.sh "base:W2S_GRAD_FP16=0" "fp16:W2S_GRAD_FP16=1" "base:W2S_GRAD_FP16=0" "fp16:W2S_GRAD_FP16=1" 2>&1 | tail -10 > gpurun_out/r3k/ab.txt
cat gpurun_out/r3k/pytest_fp16.txt gpurun_out/r3k/ab.txt; cut -c1-250 gpurun_out/r3k/fullsize_fp16.txt; python3 -c "
import json
for c in ('eog_fullsize_grad','b16_fullsize_grad'):
    d=json.load(open(f'gpurun_out/hunt/{c}.1.json')); print(c, d['worst_tensor'], d['worst_rel_l2'], len(d['over_1e3']))"
