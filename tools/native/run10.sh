mkdir -p gpurun_out/r2j
export PYTHONUNBUFFERED=1
( python -m pytest tests -m gpu -q -x ) > gpurun_out/r2j/tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2j/tests.log
tail -5 gpurun_out/r2j/tests.log
python bench.py --single-chain --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2j/fused.json
AISX_BENCH_UNFUSED=1 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2j/unfused.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2j/stats -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 > gpurun_out/r2j/prof.log 2>&1
