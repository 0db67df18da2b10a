export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d.get('roofline',{})
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r.get('kernel_ms',0),3), 'msk', round(d.get('roofline_msk',{}).get('kernel_ms',0),3))
PY
}
run() { n=$1; lib=$2; shift; shift; e=(); a=(); for w in "$@"; do case $w in --*) a+=($w);; *) e+=($w);; esac; done; env "${e[@]}" python tools/ab_bench.py $lib --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 "${a[@]}" > $O/$n.log 2>&1; ex $O/$n.log $n; }
for i in 1 2 3; do
run base_$i gr-ais_amd/lib/libaisx.so X=1
run walkprio2_$i tools/scratch/libaisx_wp2.so X=1
run walkprio1_$i tools/scratch/libaisx_wp1.so X=1
done
