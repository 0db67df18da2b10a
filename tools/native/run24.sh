mkdir -p gpurun_out/r2x
for pr in 0 1 3 0 1 3; do
python tools/ab_bench.py exp/libaisx_wprio$pr.so --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2x/wprio${pr}_$RANDOM.json
done
