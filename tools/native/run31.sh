mkdir -p gpurun_out/r3e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --single-chain --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3e/fused.json
AISX_BENCH_UNFUSED=1 python bench.py --single-chain --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r3e/unfused.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3e/fused -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > gpurun_out/r3e/prof1.log 2>&1
AISX_BENCH_UNFUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3e/unfused -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > gpurun_out/r3e/prof2.log 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r3e/tests.log
