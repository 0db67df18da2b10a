mkdir -p gpurun_out/r3v
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in mprio3 mprio0 mprio1 mprio2 mprio3 mprio0 mprio1 mprio2; do
  python tools/ab_bench.py exp/libaisx_$v.so --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v stock', d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/r3v/ab.log
done
