mkdir -p gpurun_out/r2y
for i in 1 2 3; do
python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2y/blk8_$i.json
done
python -m pytest tests/test_gpu_stages.py -m gpu -q -x 2>&1 | tail -2
