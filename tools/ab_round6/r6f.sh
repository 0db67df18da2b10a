# round 6, F: differential profile of k_corr4e_main<896> (CE_DBG builds leave parts of the tile loop out; timing only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
B=tools/native/corrbench
{
for i in 1 2; do
for v in e_best dbg1 dbg2 dbg3 dbg4 dbg8 dbg16 dbg64 dbg80 dbg83 dbg87; do
    timeout 120 $B tools/scratch/libaisx_$v.so --iters 300 2>&1 | sed 's/tags.*//'
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
