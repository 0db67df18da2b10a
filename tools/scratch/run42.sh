mkdir -p gpurun_out/r3o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export AISX_BENCH_EST_ON_PRE=1; else unset AISX_BENCH_EST_ON_PRE; fi
  python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('est_on_pre=$v stock', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/r3o/ab.log
done
export AISX_BENCH_EST_ON_PRE=1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3o/p -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > /dev/null 2>&1
