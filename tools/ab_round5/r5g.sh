# round 5, session G: how much LDS the front-end kernel should claim; generic-length correlator A/B
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5g; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d.get('roofline',{})
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r.get('kernel_ms',0),3), 'msk', round(d.get('roofline_msk',{}).get('kernel_ms',0),3), d.get('demod_ms'))
PY
}
run() { n=$1; shift; e=(); a=(); for w in "$@"; do case $w in --*) a+=($w);; *) e+=($w);; esac; done; env "${e[@]}" python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 "${a[@]}" > $O/$n.log 2>&1; ex $O/$n.log $n; }
for i in 1 2 3; do
run claim48_$i X=1
run claim56_$i AISX_AGCW_LDS_PAD=57344
run claim64_$i AISX_AGCW_LDS_PAD=65536
run claim72_$i AISX_AGCW_LDS_PAD=73728
run claim100_$i AISX_AGCW_LDS_PAD=102400
done
for i in 1 2 3; do
python bench.py --chain wideband > $O/wb_new_$i.log 2>&1; ex $O/wb_new_$i.log wb_noscratch_$i
python tools/ab_bench.py tools/scratch/libaisx_wregs.so --chain wideband > $O/wb_old_$i.log 2>&1; ex $O/wb_old_$i.log wb_spilling_$i
done
