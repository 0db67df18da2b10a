mkdir -p gpurun_out/r3w
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --single-chain --no-cpu-baseline --channels-per-gpu 8192 --steps 20 2>gpurun_out/r3w/err8192.log | tail -1 > gpurun_out/r3w/b8192.json
python bench.py --single-chain --no-cpu-baseline --channels-per-gpu 2048 2>/dev/null | tail -1 > gpurun_out/r3w/b2048.json
python bench.py --single-chain --no-cpu-baseline --chain wideband 2>gpurun_out/r3w/errwb.log | tail -1 > gpurun_out/r3w/wb.json
