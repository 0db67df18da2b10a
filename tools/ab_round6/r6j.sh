# round 6, J: differential profile of k_corr4f_main<896>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
B=tools/native/corrbench
{
for i in 1 2; do
for v in fdbg8 fdbg9 fdbg265 fdbg521 fdbg520; do
    timeout 120 $B tools/scratch/libaisx_$v.so --iters 300 2>&1 | sed 's/tags.*//'
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
