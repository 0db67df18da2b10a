mkdir -p gpurun_out/r3j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 8 16 4 8; do
  AISX_MSK_LPW=$v python bench.py --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lpw=$v stock', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_alone'])" >> gpurun_out/r3j/ab.log
done
