# round 5, session D: est with the swizzled spectrum; the recovery's CUs kept clear (AISX_MSK_LDS_PAD, KiB); corr segments
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3))
PY
}
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -m gpu > $O/pytest_stages.log 2>&1; echo "pytest stages rc=$?"; tail -3 $O/pytest_stages.log
run() { n=$1; shift; e=(); a=(); for w in "$@"; do case $w in --*) a+=($w);; *) e+=($w);; esac; done; env "${e[@]}" python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 "${a[@]}" > $O/$n.log 2>&1; ex $O/$n.log $n; }
run base0 X=1
run nseg3 AISX_CORR_NSEG=3
run nseg7 AISX_CORR_NSEG=7
run mskpad20 AISX_MSK_LDS_PAD=20
run mskpad40 AISX_MSK_LDS_PAD=40
run mskpad68 AISX_MSK_LDS_PAD=68
run mskpad20_nseg3 AISX_MSK_LDS_PAD=20 AISX_CORR_NSEG=3
run base1 X=1
run c4 X=1 --config4
run c4_nseg3 AISX_CORR_NSEG=3 --config4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/prof.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); head -9 $f | cut -c1-120
