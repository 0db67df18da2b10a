mkdir -p gpurun_out/r2l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0; do
AISX_MSK_INLINE_TAGS=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2l/inline$v -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 30 > gpurun_out/r2l/prof$v.log 2>&1
done
