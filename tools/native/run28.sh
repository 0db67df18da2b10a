mkdir -p gpurun_out/r3b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in cheapsc now; do
python tools/ab_bench.py exp/libaisx_$v.so --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 > gpurun_out/r3b/$v.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3b/stats_$v -- python tools/ab_bench.py exp/libaisx_$v.so --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > gpurun_out/r3b/prof_$v.log 2>&1
done
