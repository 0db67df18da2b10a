mkdir -p gpurun_out/r3n
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export AISX_BENCH_NO_PREPASS_WAIT=1; else unset AISX_BENCH_NO_PREPASS_WAIT; fi
  python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nowait=$v stock', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/r3n/ab.log
  python bench.py --single-chain --chain core --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nowait=$v core', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/r3n/ab.log
done
unset AISX_BENCH_NO_PREPASS_WAIT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3n/p -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > /dev/null 2>&1
