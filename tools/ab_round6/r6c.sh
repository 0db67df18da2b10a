# round 6, C: k_corr4e variants alone: stores interleaved with the butterflies (emit) / no ds_read2 merging (nols)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
B=tools/native/corrbench
D=gr-ais_amd/lib/libaisx.so
{
timeout 120 $B tools/scratch/libaisx_e.so --ref $D --iters 5
timeout 120 $B tools/scratch/libaisx_e_nols.so --ref $D --iters 5 --N 1000
for i in 1 2 3; do
  for v in "" _d_nols _e _e_noemit _e_nols _e_noemit_nols; do
    if [ -z "$v" ]; then L=$D; else L=tools/scratch/libaisx$v.so; fi
    timeout 120 $B $L --iters 20
  done
done
} > $O/log.txt 2>&1
cat $O/log.txt
