mkdir -p gpurun_out/r4l
timeout 1200 python3 tests/gpu_check.py wide wideup2 fin bwdwide wgwide > gpurun_out/r4l/check.txt 2>&1; tail -1 gpurun_out/r4l/check.txt
bash tools/ab_kbench.sh "bw64 bw64c1 bw64rd bw64u" prev bprio0 base > gpurun_out/r4l/kbench.txt 2>&1
for rep in 1 2 3; do for name in prev bprio0 base; do
  LIBENV=""; [ "$name" != base ] && LIBENV="W2S_LIB=$PWD/build_alt/libw2s_$name.so"
  env $LIBENV timeout 600 python3 bench.py --no-cpu --steps 15 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name rep $rep:', d['ms_per_step'], 'ms', d['value'], 'recordings/s')" >> gpurun_out/r4l/bench_ab.txt 2>&1
done; done
sed -n '/^columns/,$p' gpurun_out/r4l/kbench.txt | grep -v "^columns"; cat gpurun_out/r4l/bench_ab.txt
