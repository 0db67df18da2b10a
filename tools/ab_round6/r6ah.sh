# round 6, AH: the chain's walk claim (regime A) against none (AISX_WALK_LDS_PAD=0 overrides it on the experiments build)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6ah; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'msk', round(r['msk']['kernel_ms'],3), 'parity', (d.get('parity') or {}).get('bursts_identical'))
PY
}
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -2 $O/pytest_all.log
L=gr-ais_amd/lib/libaisx_exp.so
for i in 1 2 3; do
  python tools/ab_bench.py $L --no-cpu-baseline --single-chain --steps 30 > $O/d_on_$i.log 2>&1; ex $O/d_on_$i.log 4096_claim_$i
  AISX_WALK_LDS_PAD=0 python tools/ab_bench.py $L --no-cpu-baseline --single-chain --steps 30 > $O/d_off_$i.log 2>&1; ex $O/d_off_$i.log 4096_none_$i
done
for n in 2048 3072; do
  python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 --channels-per-gpu $n > $O/n${n}_on.log 2>&1; ex $O/n${n}_on.log ${n}_claim
  AISX_WALK_LDS_PAD=0 python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 --channels-per-gpu $n > $O/n${n}_off.log 2>&1; ex $O/n${n}_off.log ${n}_none
done
