mkdir -p gpurun_out/r4g
timeout 3000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r4g/pytest.txt
bash tools/ab_kbench.sh "seq1 seq32 proj qkv ff1 ff2" base wpf4 > gpurun_out/r4g/kbench_wpf4.txt 2>&1
for rep in 1 2; do for name in base wpf4; do
  LIBENV=""; [ "$name" != base ] && LIBENV="W2S_LIB=$PWD/build_alt/libw2s_$name.so"
  env $LIBENV timeout 600 python3 bench.py --no-cpu --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name rep $rep:', d['ms_per_step'], 'ms', d['value'], 'recordings/s')" >> gpurun_out/r4g/bench_wpf4.txt 2>&1
done; done
cat gpurun_out/r4g/pytest.txt; tail -30 gpurun_out/r4g/kbench_wpf4.txt; cat gpurun_out/r4g/bench_wpf4.txt
