mkdir -p gpurun_out/r3r
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stages.py -x -q 2>&1 | tail -5 > gpurun_out/r3r/tests.log
for v in late early late early; do
  AISX_BENCH_EST=$v python bench.py --single-chain --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('est=$v stock', d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['bursts_identical'], d['parity']['detections_matched_within_1'], d['msk_status'])" >> gpurun_out/r3r/ab.log
done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3r/p -- python bench.py --single-chain --no-cpu-baseline --parity-channels 0 --steps 20 > /dev/null 2>&1
