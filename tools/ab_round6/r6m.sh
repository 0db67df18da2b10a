# round 6, M: template of 112 items through the F = 4096 build (k_corr4f_main<112>, L = 3984) against the F = 2048 build
# (k_corr2d_main<112>, L = 1936); kernel trace of config 4's per-GPU shape
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6m; mkdir -p $O
B=tools/native/corrbench
X=gr-ais_amd/lib/libaisx_exp.so
{
AISX_CORR_F4=1 timeout 120 $B $X --ref gr-ais_amd/lib/libaisx.so --iters 5 --N 112
for i in 1 2 3; do
  timeout 120 $B $X --iters 300 --N 112
  AISX_CORR_F4=1 timeout 120 $B $X --iters 300 --N 112
done
for N in 64 256 400 512; do
  timeout 120 $B $X --iters 300 --N $N
  AISX_CORR_F4=1 timeout 120 $B $X --iters 300 --N $N
done
timeout 120 $B $X --iters 3000 --N 112 --nchan 256
AISX_CORR_F4=1 timeout 120 $B $X --iters 3000 --N 112 --nchan 256
} > $O/log.txt 2>&1
cat $O/log.txt | sed 's/tags.*//'
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4prof -- python bench.py --no-cpu-baseline --parity-channels 0 --single-chain --config4 --steps 30 > $O/c4prof.log 2>&1
f=$(find $O/c4prof -name '*kernel_stats.csv' | head -1); head -14 $f | cut -c1-160
