# round 5, session B: placement steering by LDS claims (does protecting the recovery's CUs pay?)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3))
PY
}
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/$n.log 2>&1; ex $O/$n.log $n
}
run base0 X=1
run corr1k AISX_CORR_LDS_PAD=1024
run agcw24k AISX_AGCW_LDS_PAD=24576
run agcw40k AISX_AGCW_LDS_PAD=40960
run agcw56k AISX_AGCW_LDS_PAD=57344
run est40k AISX_EST_LDS_PAD=40960
run agcw40k_est40k AISX_AGCW_LDS_PAD=40960 AISX_EST_LDS_PAD=40960
run all AISX_AGCW_LDS_PAD=57344 AISX_EST_LDS_PAD=57344 AISX_CORR_LDS_PAD=1024
run agcw40k_corr1k AISX_AGCW_LDS_PAD=40960 AISX_CORR_LDS_PAD=1024
run base1 X=1
# the new test of the time-parallel path
timeout 600 python -m pytest tests/test_gpu_mskp.py -x -q -m gpu -k pipelined_symbols_only > $O/pytest_mskp.log 2>&1; echo "pytest mskp rc=$?"; tail -3 $O/pytest_mskp.log
# per-kernel times of the two most promising
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -- env AISX_AGCW_LDS_PAD=40960 AISX_EST_LDS_PAD=40960 python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/prof_a.log 2>&1
f=$(find $O/prof_a -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-120
