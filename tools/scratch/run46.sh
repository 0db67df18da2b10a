mkdir -p gpurun_out/r3s
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3s/tests.log
