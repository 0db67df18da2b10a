# round 5, session H: est walking several vectors per wave; claim 72 KB default
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_chain.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d.get('roofline',{})
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r.get('kernel_ms',0),3), 'msk', round(d.get('roofline_msk',{}).get('kernel_ms',0),3))
PY
}
run() { n=$1; lib=$2; shift; shift; python tools/ab_bench.py $lib --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 "$@" > $O/$n.log 2>&1; ex $O/$n.log $n; }
for i in 1 2 3; do
run vpw4_$i gr-ais_amd/lib/libaisx.so
run vpw1_$i tools/scratch/libaisx_vpw1.so
run vpw2_$i tools/scratch/libaisx_vpw2.so
run vpw8_$i tools/scratch/libaisx_vpw8.so
run nopf_$i tools/scratch/libaisx_nopf.so
done
run c4_vpw4 gr-ais_amd/lib/libaisx.so --config4
run c4_vpw1 tools/scratch/libaisx_vpw1.so --config4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/front -- python tools/front_alone.py > $O/front.log 2>&1
f=$(find $O/front -name '*kernel_stats.csv' | head -1); head -5 $f | cut -c1-120
