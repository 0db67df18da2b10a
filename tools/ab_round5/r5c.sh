# round 5, session C: agcw v2 (peeled / ping-pong / reciprocal), est prefilter; variants
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3))
PY
}
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -m gpu > $O/pytest_stages.log 2>&1; echo "pytest stages rc=$?"; tail -3 $O/pytest_stages.log
run() { # name lib env...
  n=$1; lib=$2; shift; shift
  env "$@" python tools/ab_bench.py $lib --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/$n.log 2>&1; ex $O/$n.log $n
}
L=gr-ais_amd/lib/libaisx.so
run a0 $L X=1
run a24 $L AISX_AGCW_LDS_PAD=24576
run noslp0 tools/scratch/libaisx_noslp.so X=1
run noslp24 tools/scratch/libaisx_noslp.so AISX_AGCW_LDS_PAD=24576
run run8 tools/scratch/libaisx_run8.so X=1
run run32 tools/scratch/libaisx_run32.so X=1
run a0b $L X=1
run a24b $L AISX_AGCW_LDS_PAD=24576
run a40_corr1k $L AISX_AGCW_LDS_PAD=40960 AISX_CORR_LDS_PAD=1024
run a24_est16 $L AISX_AGCW_LDS_PAD=24576 AISX_EST_LDS_PAD=16384
python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 > $O/c4.log 2>&1; ex $O/c4.log c4
AISX_AGCW_LDS_PAD=24576 python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 > $O/c4_24.log 2>&1; ex $O/c4_24.log c4_24
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/prof.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); head -9 $f | cut -c1-120
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
