mkdir -p gpurun_out/r3p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in main own main own; do
  AISX_BENCH_EST=$v python bench.py --single-chain --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('est=$v stock', d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['bursts_identical'], d['parity']['detections_matched_within_1'])" >> gpurun_out/r3p/ab.log
done
AISX_BENCH_EST=own rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3p/p -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > /dev/null 2>&1
