# round 6, A: k_corr4e (512 threads x 8 points, <= 128 VGPRs) against k_corr4d, kernel alone (tools/native/corrbench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
B=tools/native/corrbench
D=gr-ais_amd/lib/libaisx.so
{
timeout 120 $B tools/scratch/libaisx_e.so --ref $D --iters 10
timeout 120 $B tools/scratch/libaisx_e.so --ref $D --iters 5 --N 1024
timeout 120 $B tools/scratch/libaisx_e.so --ref $D --iters 5 --N 700
for i in 1 2 3; do
  for v in "" _e _e_w3r _e_w2l; do
    if [ -z "$v" ]; then L=$D; else L=tools/scratch/libaisx$v.so; fi
    timeout 120 $B $L --iters 20
  done
done
timeout 120 $B tools/scratch/libaisx_e.so --iters 20 --nchan 256
timeout 120 $B $D --iters 20 --nchan 256
timeout 120 $B tools/scratch/libaisx_e.so --iters 20 --nchan 8192
timeout 120 $B $D --iters 20 --nchan 8192
} > $O/log.txt 2>&1
cat $O/log.txt
