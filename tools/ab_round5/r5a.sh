# round 5, session A: new streaming front-end kernel -- correctness, A/B, kernel trace, HBM ceilings
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
ex() { python - "$1" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[1], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'alone', round(r.get('kernel_ms_alone') or 0,3))
PY
}
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -m gpu > $O/pytest_stages.log 2>&1; echo "pytest stages rc=$?"; tail -3 $O/pytest_stages.log
./tools/ubench/hbm_ceiling > $O/hbm_ceiling.txt 2>&1; tail -4 $O/hbm_ceiling.txt
./tools/ubench/hbm_ceiling json > $O/hbm_ceiling.json 2>/dev/null
for i in 1 2; do
 AISX_AGC_STREAMING=0 python bench.py --no-cpu-baseline --parity-channels 0 --single-chain > $O/ab_tile_$i.log 2>&1; ex $O/ab_tile_$i.log
 python bench.py --no-cpu-baseline --parity-channels 0 --single-chain > $O/ab_stream_$i.log 2>&1; ex $O/ab_stream_$i.log
done
AISX_AGC_STREAMING=0 python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 > $O/c4_tile.log 2>&1; ex $O/c4_tile.log
python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 > $O/c4_stream.log 2>&1; ex $O/c4_stream.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/prof.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); head -14 $f | cut -c1-160
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -3 $O/pytest_all.log
