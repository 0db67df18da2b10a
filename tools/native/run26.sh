mkdir -p gpurun_out/r2z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2z/stats -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 30 > gpurun_out/r2z/prof.log 2>&1
