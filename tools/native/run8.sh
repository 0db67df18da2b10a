mkdir -p gpurun_out/r2h
for sk in none agc fs agcfs; do
  AISX_BENCH_SKIP=$sk python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2h/skip_$sk.json
done
AISX_BENCH_SKIP=agcfs python bench.py --chain core --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2h/core.json
