cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
B=tools/native/corrbench
{
for i in 1 2; do
for v in dbg8 dbg9 dbg137 dbg136; do
    timeout 120 $B tools/scratch/libaisx_$v.so --iters 300 2>&1 | sed 's/tags.*//'
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
