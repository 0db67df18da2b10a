mkdir -p gpurun_out/r2q
python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2q/unfused.json
AISX_BENCH_FUSED=1 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2q/fused_serial.json
AISX_BENCH_FUSED=1 AISX_WHATIF_SKIP_EST=1 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2q/fused_noest.json
