mkdir -p gpurun_out/r2k
export PYTHONUNBUFFERED=1
( python -m pytest tests -m gpu -q -x ) > gpurun_out/r2k/tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2k/tests.log
tail -4 gpurun_out/r2k/tests.log
for v in 1 0 1 0; do
  AISX_MSK_INLINE_TAGS=$v python bench.py --single-chain --no-cpu-baseline --parity-channels 8 2>/dev/null | tail -1 > gpurun_out/r2k/stock_inline${v}_$RANDOM.json
done
for v in 1 0; do
  AISX_MSK_INLINE_TAGS=$v python bench.py --chain core --single-chain --no-cpu-baseline --parity-channels 8 2>/dev/null | tail -1 > gpurun_out/r2k/core_inline${v}.json
done
