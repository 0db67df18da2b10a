# round 5, session F: the tree as it stands -- all gpu tests, the LDS claim the chain sets, the full default bench line
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5f; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'msk', round(d.get('roofline_msk',{}).get('kernel_ms',0),3))
PY
}
run() { n=$1; shift; e=(); a=(); for w in "$@"; do case $w in --*) a+=($w);; *) e+=($w);; esac; done; env "${e[@]}" python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 "${a[@]}" > $O/$n.log 2>&1; ex $O/$n.log $n; }
for i in 1 2; do
run claim30_$i X=1
run claim0_$i AISX_AGCW_LDS_PAD=0
run claim40_$i AISX_AGCW_LDS_PAD=40960
run claim48_$i AISX_AGCW_LDS_PAD=49152
done
run c4 X=1 --config4
run c4_claim0 AISX_AGCW_LDS_PAD=0 --config4
( time python bench.py > $O/bench_default.log 2> $O/bench_default.err ) 2> $O/bench_default.time; tail -3 $O/bench_default.time
python - <<'PY'
import json
for ln in open("gpurun_out/r5f/bench_default.log"):
    if ln.startswith('{"metric"'):
        d=json.loads(ln)
        print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"].get("achievable_GBs"), d.get("roofline_msk",{}).get("kernel_ms"), d.get("h2d"), d.get("config4_per_gpu"), d.get("config1_host_path"), d.get("parity"), d["cpu_baseline"]["reference_volk_path"][:80])
        print([ (c["channels"],c["template_len"],round(c["frac"],3)) for c in d.get("corr_only",[])])
PY
python bench.py --chain wideband > $O/wideband.log 2>&1; tail -c 600 $O/wideband.log
