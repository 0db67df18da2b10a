mkdir -p gpurun_out/r3q
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base est2 est2v; do
 for m in main own; do
  AISX_BENCH_EST=$m python tools/ab_bench.py exp/libaisx_$v.so --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v est=$m stock', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/r3q/ab.log
 done
done
AISX_BENCH_EST=own rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3q/p -- python tools/ab_bench.py exp/libaisx_est2v.so --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > /dev/null 2>&1
