mkdir -p gpurun_out/r3m
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 2 0 2; do
  AISX_BENCH_PRIO=$v python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio=$v stock', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/r3m/ab.log
done
AISX_BENCH_PRIO=2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3m/p2 -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > /dev/null 2>&1
