# round 6, T: the resolver on its own stream where the sample passes bound the step (AISX_CHAIN_RES_STREAM=0/1 on the
# experiments build; the product's rule: more recovery workgroups than half the CUs); resolver waves per channel by channel count
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6t; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'msk', round(r['msk']['kernel_ms'],3), 'parity', (d.get('parity') or {}).get('bursts_identical'), 'status', d.get('msk_status'))
PY
}
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
L=gr-ais_amd/lib/libaisx_exp.so
for i in 1 2; do
for rs in 0 1; do
  AISX_CHAIN_RES_STREAM=$rs python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $O/c4_rs${rs}_$i.log 2>&1; ex $O/c4_rs${rs}_$i.log c4_rs${rs}_$i
  AISX_CHAIN_RES_STREAM=$rs python tools/ab_bench.py $L --no-cpu-baseline --single-chain --steps 30 > $O/d_rs${rs}_$i.log 2>&1; ex $O/d_rs${rs}_$i.log d_rs${rs}_$i
done
done
for n in 5120 6144 12288; do
for rs in 0 1; do
  AISX_CHAIN_RES_STREAM=$rs python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --channels-per-gpu $n --steps 20 > $O/n${n}_rs${rs}.log 2>&1; ex $O/n${n}_rs${rs}.log n${n}_rs${rs}
done
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4prof -- python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $O/c4prof.log 2>&1
f=$(find $O/c4prof -name '*kernel_stats.csv' | head -1); head -10 $f | cut -c1-150
