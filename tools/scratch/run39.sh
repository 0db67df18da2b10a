mkdir -p gpurun_out/r3l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3l/tests.log
python bench.py --no-cpu-baseline --single-chain 2>/dev/null | tail -1 > gpurun_out/r3l/bench.json
