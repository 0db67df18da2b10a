# round 6, Q: 16 channels per recovery wave at 8192 channels (128 workgroups on half of the CUs instead of 256 on all)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6q; mkdir -p $O
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
for lpw in 8 16 32; do
  AISX_MSK_LPW=$lpw python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $O/c4_${lpw}_$i.log 2>&1; ex $O/c4_${lpw}_$i.log c4_lpw${lpw}_$i
done
done
AISX_MSK_LPW=16 python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --channels-per-gpu 16384 --steps 20 > $O/c16k_16.log 2>&1; ex $O/c16k_16.log 16384_lpw16
AISX_MSK_LPW=8 python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --channels-per-gpu 16384 --steps 20 > $O/c16k_8.log 2>&1; ex $O/c16k_8.log 16384_lpw8
tail -3 $O/c16k_8.log | cut -c1-300
