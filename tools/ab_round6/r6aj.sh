# round 6, AJ: the bit tail's and the resolver's workgroups kept off the recovery's CUs by LDS claims (4096 channels)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6aj; mkdir -p $O
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
  python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/base_$i.log 2>&1; ex $O/base_$i.log base_$i
  AISX_TAIL_LDS_PAD=73728 python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/tail_$i.log 2>&1; ex $O/tail_$i.log tail72k_$i
  AISX_RSV_LDS_PAD=65536 python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/rsv_$i.log 2>&1; ex $O/rsv_$i.log rsv64k_$i
  AISX_TAIL_LDS_PAD=73728 AISX_RSV_LDS_PAD=65536 python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/both_$i.log 2>&1; ex $O/both_$i.log both_$i
done
