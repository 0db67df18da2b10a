mkdir -p gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3g/tests.log
for v in base stage base stage; do
  python tools/ab_bench.py exp/libaisx_$v.so --single-chain --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v stock', d['ms_per_step'])" >> gpurun_out/r3g/ab.log
  python tools/ab_bench.py exp/libaisx_$v.so --single-chain --chain core --no-cpu-baseline --parity-channels 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v core', d['ms_per_step'])" >> gpurun_out/r3g/ab.log
done
