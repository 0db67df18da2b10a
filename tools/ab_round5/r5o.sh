export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5o; mkdir -p $O
python bench.py --no-cpu-baseline --parity-channels 0 --steps 30 > $O/full.log 2>/dev/null
python - $O/full.log <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print('default', round(d['ms_per_step'],3), 'corr', round(r['kernel_ms'],3), 'msk', round(d['roofline_msk']['kernel_ms'],3))
        print('no lookahead', round(d['no_lookahead_ms_per_step'],3), 'corr', round(d['no_lookahead_corr_kernel_ms'],3), 'msk', round(d.get('no_lookahead_msk_kernel_ms',0),3))
        c=d['corr_est_to_msk_only']; print('core chain', round(c['ms_per_step'],3), 'corr', round(c['corr_kernel_ms'],3))
PY
