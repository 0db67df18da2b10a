mkdir -p gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base nostore base nostore; do
  python tools/ab_bench.py exp/libaisx_$v.so --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v stock', d['ms_per_step'])" >> gpurun_out/r3f/ab.log
  python tools/ab_bench.py exp/libaisx_$v.so --single-chain --chain core --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v core', d['ms_per_step'])" >> gpurun_out/r3f/ab.log
done
