# round 6, I: k_corr4f (next window in registers, one image, two barriers) against 4e / 4d, steady state
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
B=tools/native/corrbench
D=gr-ais_amd/lib/libaisx.so
{
timeout 120 $B tools/scratch/libaisx_f.so --ref $D --iters 5
timeout 120 $B tools/scratch/libaisx_f.so --ref $D --iters 5 --N 1000
timeout 120 $B tools/scratch/libaisx_f.so --ref $D --iters 5 --N 2047
timeout 120 $B tools/scratch/libaisx_f.so --ref $D --iters 5 --N 513 --n 40000 --nchan 300
for i in 1 2 3; do
  for v in "" _e_best _f _f_w2l _f_ls _f_emit0; do
    if [ -z "$v" ]; then L=$D; else L=tools/scratch/libaisx$v.so; fi
    timeout 120 $B $L --iters 300
  done
done
for v in "" _f; do
    if [ -z "$v" ]; then L=$D; else L=tools/scratch/libaisx$v.so; fi
    timeout 120 $B $L --iters 300 --N 1024
    timeout 120 $B $L --iters 300 --N 700
    timeout 120 $B $L --iters 300 --nchan 8192
    timeout 120 $B $L --iters 3000 --nchan 256
done
} > $O/log.txt 2>&1
cat $O/log.txt
