mkdir -p gpurun_out/r2u
for pr in 0 1 2; do
python tools/ab_bench.py exp/libaisx_prio$pr.so --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2u/fused_prio$pr.json
AISX_BENCH_UNFUSED=1 python tools/ab_bench.py exp/libaisx_prio$pr.so --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2u/unfused_prio$pr.json
done
