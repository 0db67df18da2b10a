mkdir -p gpurun_out/r2e
export PYTHONUNBUFFERED=1
( time python -m pytest tests -m gpu -q -x --durations=8 ) > gpurun_out/r2e/tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2e/tests.log
( time python bench.py ) > gpurun_out/r2e/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r2e/bench.log
AISX_CORR_DMA=0 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 > gpurun_out/r2e/bench_nodma.log 2>&1
tail -4 gpurun_out/r2e/tests.log
