# round 6, AI: the estimates kept off the recovery's CUs too (AISX_EST_LDS_PAD), now that the sample passes have slack at 4096 channels
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ai; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'msk', round(r['msk']['kernel_ms'],3))
PY
}
L=gr-ais_amd/lib/libaisx_exp.so
for i in 1 2 3; do
for pad in 0 40960 20480; do
  AISX_EST_LDS_PAD=$pad python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/d_${pad}_$i.log 2>&1; ex $O/d_${pad}_$i.log est_pad${pad}_$i
done
done
