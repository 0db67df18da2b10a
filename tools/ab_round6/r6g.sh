# round 6, G: differential profile, second set
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6g; mkdir -p $O
B=tools/native/corrbench
{
for i in 1 2; do
for v in e_best dbg8 dbg9 dbg10 dbg11 dbg15 dbg27 dbg75 dbg79; do
    timeout 120 $B tools/scratch/libaisx_$v.so --iters 300 2>&1 | sed 's/tags.*//'
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
