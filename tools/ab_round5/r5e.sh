# round 5, session E: front-end kernels alone; placement candidates, three interleaved repetitions each
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/front -- python tools/scratch/front_alone.py > $O/front.log 2>&1
f=$(find $O/front -name '*kernel_stats.csv' | head -1); head -6 $f | cut -c1-120
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'msk', round(d.get('roofline_msk',{}).get('kernel_ms',0),3))
PY
}
run() { n=$1; shift; e=(); a=(); for w in "$@"; do case $w in --*) a+=($w);; *) e+=($w);; esac; done; env "${e[@]}" python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 "${a[@]}" > $O/$n.log 2>&1; ex $O/$n.log $n; }
for i in 1 2 3; do
run base_$i X=1
run mskpad40_$i AISX_MSK_LDS_PAD=40
run agcw40_$i AISX_AGCW_LDS_PAD=40960
run agcw40corr1_$i AISX_AGCW_LDS_PAD=40960 AISX_CORR_LDS_PAD=1024
run mskpad40nseg3_$i AISX_MSK_LDS_PAD=40 AISX_CORR_NSEG=3
done
