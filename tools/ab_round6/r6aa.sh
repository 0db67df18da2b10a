# round 6, AA: 4 channels per recovery wave (256 workgroups at 4096 channels) with round 6's correlator and claim rule
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6aa; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'msk', round(r['msk']['kernel_ms'],3), 'status', d.get('msk_status'))
PY
}
L=gr-ais_amd/lib/libaisx_exp.so
for i in 1 2; do
for lpw in 8 4; do
  AISX_MSK_LPW=$lpw python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/d_${lpw}_$i.log 2>&1; ex $O/d_${lpw}_$i.log d_lpw${lpw}_$i
  AISX_MSK_LPW=$lpw python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 --channels-per-gpu 2048 > $O/h_${lpw}_$i.log 2>&1; ex $O/h_${lpw}_$i.log 2048_lpw${lpw}_$i
done
done
