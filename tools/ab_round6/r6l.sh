# round 6, L: the tree with k_corr4f as the product's correlator: all -m gpu tests, the default bench line (three runs), config 4's shape
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'alone', round(r.get('kernel_ms_alone') or 0,3), 'frac', round(r['frac'],3))
PY
}
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
for i in 1 2 3; do
  python bench.py > $O/bench_$i.log 2>&1; ex $O/bench_$i.log default_$i
done
python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 > $O/c4.log 2>&1; ex $O/c4.log c4
tail -1 $O/bench_3.log
