mkdir -p gpurun_out/r2b
export PYTHONUNBUFFERED=1
( time python -m pytest tests -m gpu -q -s -x --durations=15 ) > gpurun_out/r2b/tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2b/tests.log
( time python bench.py --steps 20 ) > gpurun_out/r2b/bench.log 2>&1
echo "bench rc=$?" >> gpurun_out/r2b/bench.log
tail -5 gpurun_out/r2b/tests.log; tail -3 gpurun_out/r2b/bench.log
