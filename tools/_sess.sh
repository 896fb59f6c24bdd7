mkdir -p gpurun_out/r4m
bash tools/ab_kbench.sh "b16 b16u b32 b32u b21 bfirst bw64 d128 u128" erfhi base > gpurun_out/r4m/kbench.txt 2>&1
for rep in 1 2 3; do for name in erfhi base; do
  LIBENV=""; [ "$name" != base ] && LIBENV="W2S_LIB=$PWD/build_alt/libw2s_$name.so"
  env $LIBENV timeout 600 python3 bench.py --no-cpu --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name rep $rep:', d['ms_per_step'], 'ms', d['value'], 'recordings/s')" >> gpurun_out/r4m/bench_ab.txt 2>&1
done; done
timeout 2400 python3 -m pytest tests -m gpu -x -q -k "gradient or kernels or train or golden or stock or accum" 2>&1 | tail -4 > gpurun_out/r4m/pytest.txt
sed -n '/^== /,$p' gpurun_out/r4m/kbench.txt; cat gpurun_out/r4m/bench_ab.txt gpurun_out/r4m/pytest.txt
