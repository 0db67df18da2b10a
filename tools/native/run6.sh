mkdir -p gpurun_out/r2f
export PYTHONUNBUFFERED=1
for lpw in 8 4 8 4; do
  AISX_MSK_LPW=$lpw python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2f/stock_lpw${lpw}_$RANDOM.json
done
for lpw in 8 4; do
  AISX_MSK_LPW=$lpw python bench.py --chain core --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2f/core_lpw${lpw}.json
done
python -m pytest tests/test_gpu_corr_msk.py -m gpu -q -x -k "msk or corr_dense" 2>&1 | tail -3 > gpurun_out/r2f/tests.log
cat gpurun_out/r2f/tests.log
