# round 6, AF: aisx_corr_set_lds_claim: its test, and the default line with the new side measurement
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6af; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_corr_msk.py -x -q -m gpu -k "claim or dense_matches" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench.log 2>&1
python - <<'PY'
import json
for ln in open('gpurun_out/r6af/bench.log'):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); s=d['config']['side']
        print('default', d['ms_per_step'], d['value'], 'corr', d['roofline']['kernel_ms'], 'msk', d['roofline']['msk']['kernel_ms'])
        print('corr_off_recovery_cus', s.get('corr_off_recovery_cus'))
PY
tail -2 $O/bench.log | cut -c1-300
