# round 6, E: as C, in the steady state (300 launches, the library keeps the last 64 durations: clocks have ramped up)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
B=tools/native/corrbench
D=gr-ais_amd/lib/libaisx.so
{
for i in 1 2 3; do
  for v in "" _d_nols _e _e_noemit _e_nols _e_noemit_nols; do
    if [ -z "$v" ]; then L=$D; else L=tools/scratch/libaisx$v.so; fi
    timeout 120 $B $L --iters 300
  done
done
for v in "" _e_noemit_nols; do
    if [ -z "$v" ]; then L=$D; else L=tools/scratch/libaisx$v.so; fi
    timeout 120 $B $L --iters 300 --N 1024
    timeout 120 $B $L --iters 300 --N 700
    timeout 120 $B $L --iters 300 --nchan 8192
    timeout 120 $B $L --iters 3000 --nchan 256
    timeout 120 $B $L --iters 300 --N 112
done
} > $O/log.txt 2>&1
cat $O/log.txt
