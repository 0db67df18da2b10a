mkdir -p gpurun_out/r3c
export PYTHONUNBUFFERED=1
( python -m pytest tests -m gpu -q -x ) > gpurun_out/r3c/tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3c/tests.log
tail -4 gpurun_out/r3c/tests.log
for i in 1 2; do
python bench.py --single-chain --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3c/fused_$i.json
AISX_BENCH_UNFUSED=1 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r3c/unfused_$i.json
done
