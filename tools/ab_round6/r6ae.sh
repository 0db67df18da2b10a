# round 6, AE: the chain in view mode (corr_est writes no pass-through; the recovery reads the delayed stream from the front end's
# rotating buffers): all -m gpu tests, default line and config 4's shape against AISX_CHAIN_VIEW=0 on the experiments build
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ae; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'alone', round(r['kernel_ms_alone'],3), 'msk', round(r['msk']['kernel_ms'],3), 'parity', (d.get('parity') or {}).get('bursts_identical'), 'status', d.get('msk_status'))
PY
}
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
L=gr-ais_amd/lib/libaisx_exp.so
for i in 1 2; do
for v in 1 0; do
  AISX_CHAIN_VIEW=$v python tools/ab_bench.py $L --no-cpu-baseline --single-chain --steps 30 > $O/d_v${v}_$i.log 2>&1; ex $O/d_v${v}_$i.log d_view${v}_$i
  AISX_CHAIN_VIEW=$v python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $O/c4_v${v}_$i.log 2>&1; ex $O/c4_v${v}_$i.log c4_view${v}_$i
done
done
