export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2 3; do
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_$i.log 2> $O/driver_$i.err ) 2> $O/driver_$i.time
python - $O/driver_$i.log <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print('driver-style', round(d['value']), round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'alone', round(r['frac_alone'],3), 'msk', round(d['roofline_msk']['kernel_ms'],3), 'c4', round(d['config4_per_gpu']['ms_per_step'],3), [round(c['frac'],3) for c in d['corr_only']])
PY
grep real $O/driver_$i.time
done
