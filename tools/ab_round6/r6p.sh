# round 6, P: k_fs_est with 2 / 1 waves per workgroup (18 / 9 KB of LDS instead of 35: more than one workgroup fits a CU beside
# a timing-recovery workgroup) against the 4-wave build, 8192 and 4096 channels
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6p; mkdir -p $O
ex() { python - "$1" "$2" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{"metric"'):
        d=json.loads(ln); r=d['roofline']
        print(sys.argv[2], 'ms/step', round(d['ms_per_step'],3), 'corr in-chain', round(r['kernel_ms'],3), 'msk', round(r['msk']['kernel_ms'],3))
PY
}
for i in 1 2; do
for v in exp est2 est1; do
  if [ $v = exp ]; then L=gr-ais_amd/lib/libaisx_exp.so; else L=tools/scratch/libaisx_$v.so; fi
  python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $O/c4_${v}_$i.log 2>&1; ex $O/c4_${v}_$i.log c4_${v}_$i
  python tools/ab_bench.py $L --no-cpu-baseline --parity-channels 0 --single-chain --steps 30 > $O/d_${v}_$i.log 2>&1; ex $O/d_${v}_$i.log d_${v}_$i
done
done
