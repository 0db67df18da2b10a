mkdir -p gpurun_out/r2i
for sk in none agc+pre fs+agc+pre; do
  AISX_BENCH_SKIP=$sk python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2i/skip_$sk.json
done
