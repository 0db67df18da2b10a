export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5n; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
for i in 1 2; do python bench.py --chain wideband 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wideband', round(d['ms_per_step'],3), round(d['pfb_ms'],3), round(d['demod_ms'],3))"; done
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.log 2>/dev/null
python - $O/driver.log <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print('driver-style', round(d['value']), round(d['ms_per_step'],3), 'frac', round(r['frac'],3), 'alone', round(r['frac_alone'],3), 'msk', round(d['roofline_msk']['kernel_ms'],3), 'c4', round(d['config4_per_gpu']['ms_per_step'],3), 'c5', round(d['config5_wideband']['ms_per_step'],3), [round(c['frac'],3) for c in d['corr_only']])
PY
