mkdir -p gpurun_out/r4j
timeout 1200 python3 tests/gpu_check.py wide wideup2 fin bwdwide wgwide > gpurun_out/r4j/check.txt 2>&1; tail -2 gpurun_out/r4j/check.txt
bash tools/ab_kbench.sh "f128 f64 f128s2 f64s2 f3264 f64128 d128 d64 u128 u64" wold base > gpurun_out/r4j/kbench.txt 2>&1
for rep in 1 2; do for name in wold base; do
  LIBENV=""; [ "$name" != base ] && LIBENV="W2S_LIB=$PWD/build_alt/libw2s_$name.so"
  env $LIBENV timeout 600 python3 bench.py --no-cpu --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name rep $rep:', d['ms_per_step'], 'ms', d['value'], 'recordings/s')" >> gpurun_out/r4j/bench_ab.txt 2>&1
done; done
sed -n '/^columns/,$p' gpurun_out/r4j/kbench.txt; cat gpurun_out/r4j/bench_ab.txt
