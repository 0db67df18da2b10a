mkdir -p gpurun_out/r2t
for i in 1 2; do
python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2t/fused_$i.json
AISX_BENCH_UNFUSED=1 python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r2t/unfused_$i.json
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2t/stats -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 30 > gpurun_out/r2t/prof.log 2>&1
