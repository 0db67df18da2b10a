mkdir -p gpurun_out/r2s
python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2s/fused_1.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2s/stats -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 30 > gpurun_out/r2s/prof.log 2>&1
