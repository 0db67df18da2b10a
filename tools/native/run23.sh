mkdir -p gpurun_out/r2w
for i in 1 2; do
python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2w/fused_split_$i.json
AISX_BENCH_EST_ON_PRE=1 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2w/fused_pre_$i.json
done
python -m pytest tests/test_gpu_stages.py -m gpu -q -x 2>&1 | tail -2
