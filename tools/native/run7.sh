mkdir -p gpurun_out/r2g
export PYTHONUNBUFFERED=1
( time python -m pytest tests -m gpu -q -x -s --durations=5 ) > gpurun_out/r2g/tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2g/tests.log
grep -v "^\s*$" gpurun_out/r2g/tests.log | tail -25
