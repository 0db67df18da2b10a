mkdir -p gpurun_out/r3h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3h/tests.log
python bench.py 2>gpurun_out/r3h/bench.err | tail -1 > gpurun_out/r3h/bench.json
