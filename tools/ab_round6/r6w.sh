# round 6, W: the 512-thread correlator kept off the CUs that hold a recovery workgroup (AISX_CORR_LDS_PAD: 55 + 17 KB does not fit beside 90)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6w; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'msk', round(r['msk']['kernel_ms'],3))
PY
}
L=gr-ais_amd/lib/libaisx_exp.so
for i in 1 2; do
for pad in 0 6144 17408; do
  AISX_CORR_LDS_PAD=$pad python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/d_${pad}_$i.log 2>&1; ex $O/d_${pad}_$i.log d_pad${pad}_$i
done
done
