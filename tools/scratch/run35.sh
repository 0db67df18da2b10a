mkdir -p gpurun_out/r3i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  AISX_CORR_DMA=$v python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dma=$v stock', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_alone'])" >> gpurun_out/r3i/ab.log
done
