# round 6, S: the resolver on sixteen waves per channel: -m gpu tests, default line, config 4's shape, kernel trace
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'alone', round(r.get('kernel_ms_alone') or 0,3), 'msk', round(r['msk']['kernel_ms'],3), 'parity', (d.get('parity') or {}).get('bursts_identical'))
PY
}
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
for i in 1 2; do
  python bench.py --no-cpu-baseline --single-chain > $O/d_$i.log 2>&1; ex $O/d_$i.log default_$i
  python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 > $O/c4_$i.log 2>&1; ex $O/c4_$i.log c4_$i
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4prof -- python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $O/c4prof.log 2>&1
f=$(find $O/c4prof -name '*kernel_stats.csv' | head -1); head -12 $f | cut -c1-150
