mkdir -p gpurun_out/r3d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/fused -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > gpurun_out/r3d/prof1.log 2>&1
AISX_BENCH_UNFUSED=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d/unfused -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > gpurun_out/r3d/prof2.log 2>&1
