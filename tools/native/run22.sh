mkdir -p gpurun_out/r2v
for pr in 1 0 1 0; do
AISX_BENCH_PRIO=$pr python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2v/fused_sprio${pr}_$RANDOM.json
done
AISX_BENCH_PRIO=1 AISX_BENCH_UNFUSED=1 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2v/unfused_sprio1.json
