# round 6, AD: what the step would gain if the correlator did not write its pass-through at all (CE_DBG=1 build: the stores
# left out, timing only -- the recovery then reads an unwritten buffer): 8192 and 4096 channels
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ad; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'alone', round(r['kernel_ms_alone'],3), 'msk', round(r['msk']['kernel_ms'],3), 'status', d.get('msk_status'))
PY
}
for i in 1 2; do
for v in exp nost; do
  if [ $v = exp ]; then L=gr-ais_amd/lib/libaisx_exp.so; else L=tools/scratch/libaisx_$v.so; fi
  python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $O/c4_${v}_$i.log 2>&1; ex $O/c4_${v}_$i.log c4_${v}_$i
  python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/d_${v}_$i.log 2>&1; ex $O/d_${v}_$i.log d_${v}_$i
done
done
